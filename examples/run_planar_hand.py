#!/usr/bin/env python3
"""The reference's planar-hand scripts on the GPU:

    python examples/run_planar_hand.py irs_lqr              # examples/planar_hand/run_planar_hand.py
    python examples/run_planar_hand.py irs_lqr --bounds rel # u_bounds_rel instead of the trust region
    python examples/run_planar_hand.py cem                  # examples/planar_hand/run_planar_hand_cem.py

Problem data as in run_planar_hand.py:20-153 (horizon 3 s at h = 0.1, initial grasp, goal
q_u0 + (0.3, -0.1, 0.5), Q/Qd/R dicts, u_bounds_abs = +-0.5 h, std_u_initial = 0.3 / iter^0.8,
20 iterations) with two differences forced by the missing simulator: the contact step is the device
functor (parity unpinned) and gradient_mode is "zero_order_B" (the script's
"first_order" needs the simulator's derivatives); --N defaults to 1000 samples instead of 50.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irs_mpc_amd as amd  # noqa: E402


def problem(T, h=0.1):
    q_dynamics = amd.PlanarHandDynamics(h)
    idx_u, idx_a_l, idx_a_r = "sphere", "arm_left", "arm_right"
    q_u0 = np.array([0.0, 0.35, 0.0])
    qa_l, qa_r = np.array([-np.pi / 4, -np.pi / 4]), np.array([np.pi / 4, np.pi / 4])
    x0 = q_dynamics.get_x_from_q_dict({idx_u: q_u0, idx_a_l: qa_l, idx_a_r: qa_r})
    u_traj_0 = np.tile(q_dynamics.get_u_from_q_cmd_dict({idx_a_l: qa_l, idx_a_r: qa_r}), (T, 1))
    Q_dict = {idx_u: np.array([1e-3, 1e-3, 10]), idx_a_l: np.array([1e-3, 1e-3]), idx_a_r: np.array([1e-3, 1e-3])}
    Qd_dict = {model: Q_i * 100 for model, Q_i in Q_dict.items()}
    R_dict = {idx_a_l: 5 * np.array([1, 1]), idx_a_r: 5 * np.array([1, 1])}
    xd = q_dynamics.get_x_from_q_dict({idx_u: q_u0 + np.array([0.3, -0.1, 0.5]), idx_a_l: qa_l, idx_a_r: qa_r})
    return q_dynamics, x0, u_traj_0, Q_dict, Qd_dict, R_dict, np.tile(xd, (T + 1, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("method", choices=["irs_lqr", "cem"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--T", type=int, default=30)
    ap.add_argument("--N", type=int, default=1000, help="samples per timestep / CEM batch size")
    ap.add_argument("--bounds", choices=["abs", "rel", "none"], default="abs")
    ap.add_argument("--device-rng", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--csv", default=None)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()

    h = 0.1
    q_dynamics, x0, u_traj_0, Q_dict, Qd_dict, R_dict, x_trj_d = problem(a.T, h)
    np.random.seed(a.seed)
    if a.method == "irs_lqr":
        params = amd.IrsLqrQuasistaticParameters()
        params.Q_dict, params.Qd_dict, params.R_dict = Q_dict, Qd_dict, R_dict
        params.x0, params.x_trj_d, params.u_trj_0, params.T = x0, x_trj_d, u_traj_0, a.T
        dim_u = q_dynamics.dim_u
        if a.bounds == "abs":       # run_planar_hand.py:138-139
            params.u_bounds_abs = np.array([-np.ones(dim_u) * 0.5 * h, np.ones(dim_u) * 0.5 * h])
        elif a.bounds == "rel":     # e.g. examples/box_pushing/run_box_pushing.py:117
            params.u_bounds_rel = np.array([-np.ones(dim_u) * 0.3 * h, np.ones(dim_u) * 0.3 * h])
        params.sampling = lambda u_initial, it: u_initial / (it ** 0.8)     # :142-143
        params.std_u_initial = np.ones(dim_u) * 0.3
        params.num_samples = a.N
        params.publish_every_iteration = False
        if a.device_rng:
            params.device_rng_seed = a.seed
        solver = amd.IrsLqrQuasistatic(q_dynamics=q_dynamics, params=params)
    else:
        params = amd.CemQuasistaticParameters()
        params.Q_dict, params.Qd_dict, params.R_dict = Q_dict, Qd_dict, R_dict
        params.x0, params.xd_trj, params.u_trj_0, params.T = x0, x_trj_d, u_traj_0, a.T
        params.n_elite, params.batch_size = max(2, a.N // 10), a.N
        params.initial_std = 0.1 * np.ones(q_dynamics.dim_u)
        params.publish_every_iteration = False
        solver = amd.CrossEntropyMethodQuasistatic(q_dynamics, params)
    solver.verbose = not a.quiet
    t0 = time.time()
    solver.iterate(a.iters)
    print("Final cost: " + str(solver.cost) + "  best: " + str(solver.cost_best))
    print("Elapsed time: " + str(time.time() - t0))
    print("cost history:", " ".join("%.6f" % c for c in solver.cost_all_list))
    if a.csv:
        np.savetxt(a.csv, np.array(solver.cost_all_list), delimiter=",")


if __name__ == "__main__":
    main()
