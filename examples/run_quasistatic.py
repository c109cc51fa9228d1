#!/usr/bin/env python3
"""The reference's quasistatic contact scripts on the GPU:

    python examples/run_quasistatic.py planar_hand irs_lqr       # examples/planar_hand/run_planar_hand.py
    python examples/run_quasistatic.py planar_hand irs_lqr --bounds rel
    python examples/run_quasistatic.py planar_hand cem           # .../run_planar_hand_cem.py
    python examples/run_quasistatic.py box_pivoting irs_lqr      # examples/box_pivoting/run_box_pivoting.py
    python examples/run_quasistatic.py box_pivoting cem          # .../run_box_pivoting_cem.py
    python examples/run_quasistatic.py box_pushing irs_lqr       # examples/box_pushing/run_box_pushing.py
    python examples/run_quasistatic.py planar_hand_spin irs_lqr  # examples/planar_hand/run_planar_hand_spin.py

planar_hand: problem data as in run_planar_hand.py:20-153 (h = 0.1, initial grasp, goal
q_u0 + (0.3, -0.1, 0.5), Q/Qd/R dicts, u_bounds_abs = +-0.5 h, std_u_initial = 0.3 / iter^0.8).
box_pivoting: run_box_pivoting.py:20-131 (hand sweeps from (-0.5, 0.5) to (0.5, 0.5), goal box pose
q_u0 + (1, 0.5, -pi/2), Q = (5, 5, 50 | 0, 0), R = 1e3, u_bounds_rel = +-0.15 h, std 0.1^(0.5 iter)).
One difference forced by the missing simulator: the contact step is the device functor (pinned
for box_pushing, unpinned geometry for the others).  gradient_mode defaults to the set-up files'
(planar_hand_setup.py:28 "first_order" -- the step's active-set derivative, computed per sample on the
device; "zero_order_B" for the boxes), --gradient-mode overrides; --N defaults to 1000 samples instead of
50 / 100.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irs_mpc_amd as amd  # noqa: E402


def box_problem(T, h=0.1):
    q_dynamics = amd.BoxPivotingDynamics(h)
    idx_u, idx_a = "box", "hand"
    q_u0 = np.array([0.0, 0.5, 0.0])
    qa0, qa1 = np.array([-0.5, 0.5]), np.array([0.5, 0.5])
    x0 = q_dynamics.get_x_from_q_dict({idx_u: q_u0, idx_a: qa0})
    # FirstOrderHold from qa0 to qa1, sampled at t + h (run_box_pivoting.py:24-26, :83-87)
    u_traj_0 = np.stack([qa0 + (qa1 - qa0) * (t + 1) / T for t in range(T)])
    Q_dict = {idx_u: np.array([5, 5, 50]), idx_a: np.array([0.0, 0.0])}
    Qd_dict = {model: Q_i * 1 for model, Q_i in Q_dict.items()}
    R_dict = {idx_a: 1e3 * np.array([1, 1])}
    xd = q_dynamics.get_x_from_q_dict({idx_u: q_u0 + np.array([1.0, 0.5, -np.pi / 2]), idx_a: qa0})
    return q_dynamics, x0, u_traj_0, Q_dict, Qd_dict, R_dict, np.tile(xd, (T + 1, 1))


def push_problem(T, h=0.1):
    """examples/box_pushing/run_box_pushing.py:20-131."""
    q_dynamics = amd.BoxPushingDynamics(h)
    idx_u, idx_a = "box", "hand"
    q_u0, qa0 = np.array([0.0, 0.5, 0.0]), np.array([0.0, -0.2])
    x0 = q_dynamics.get_x_from_q_dict({idx_u: q_u0, idx_a: qa0})
    u_traj_0 = np.tile(qa0, (T, 1))                               # :24-27 (hold)
    Q_dict = {idx_u: np.array([3.0, 3.0, 1.2]), idx_a: np.array([0.0, 0.0])}            # :101-103
    Qd_dict = {model: Q_i * 0 for model, Q_i in Q_dict.items()}                        # :104
    R_dict = {idx_a: 1e1 * np.array([1, 1])}                                           # :105
    xd = q_dynamics.get_x_from_q_dict({idx_u: q_u0 + np.array([0.5, 0.5, -np.pi / 4]), idx_a: qa0})   # :107-109
    return q_dynamics, x0, u_traj_0, Q_dict, Qd_dict, R_dict, np.tile(xd, (T + 1, 1))


def spin_problem(T, h=0.1):
    """examples/planar_hand/run_planar_hand_spin.py:33-132: the disc held higher in a wider grasp, to be
    lowered 0.2 and spun by -pi/4; x/theta weighted 10, Qd = 10 Q, R = 100."""
    q_dynamics = amd.PlanarHandDynamics(h)
    idx_u, idx_a_l, idx_a_r = "sphere", "arm_left", "arm_right"
    q_u0 = np.array([0.0, 0.6, 0.0])
    qa_l, qa_r = np.array([-np.pi / 2 + 0.5, -np.pi / 2 + 0.5]), np.array([np.pi / 2 - 0.5, np.pi / 2 - 0.5])
    x0 = q_dynamics.get_x_from_q_dict({idx_u: q_u0, idx_a_l: qa_l, idx_a_r: qa_r})
    u_traj_0 = np.tile(q_dynamics.get_u_from_q_cmd_dict({idx_a_l: qa_l, idx_a_r: qa_r}), (T, 1))
    Q_dict = {idx_u: np.array([10.0, 1.0, 10.0]), idx_a_l: np.array([1e-3, 1e-3]), idx_a_r: np.array([1e-3, 1e-3])}
    Qd_dict = {model: Q_i * 10 for model, Q_i in Q_dict.items()}
    R_dict = {idx_a_l: 1e2 * np.array([1, 1]), idx_a_r: 1e2 * np.array([1, 1])}
    xd = q_dynamics.get_x_from_q_dict({idx_u: q_u0 + np.array([0.0, -0.2, -np.pi / 4]), idx_a_l: qa_l, idx_a_r: qa_r})
    return q_dynamics, x0, u_traj_0, Q_dict, Qd_dict, R_dict, np.tile(xd, (T + 1, 1))


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
    ap.add_argument("system", choices=["planar_hand", "planar_hand_spin", "box_pivoting", "box_pushing"])
    ap.add_argument("method", choices=["irs_lqr", "cem"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--T", type=int, default=30)
    ap.add_argument("--N", type=int, default=1000, help="samples per timestep / CEM batch size")
    ap.add_argument("--bounds", choices=["abs", "rel", "none"], default=None,
                    help="default: the script's own (planar_hand: abs, box_pivoting: rel)")
    ap.add_argument("--gradient-mode", default=None, choices=["zero_order_B", "zero_order_AB", "first_order", "exact"],
                    help="default: the reference's set-up files (planar_hand_setup.py:28 first_order; "
                         "box_pivoting_setup.py / box_pushing_setup.py:25 zero_order_B)")
    ap.add_argument("--device-rng", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--csv", default=None)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()

    h = 0.1
    hand = a.system in ("planar_hand", "planar_hand_spin")
    spin = a.system == "planar_hand_spin"
    make = {"planar_hand": problem, "planar_hand_spin": spin_problem, "box_pivoting": box_problem,
            "box_pushing": push_problem}[a.system]
    q_dynamics, x0, u_traj_0, Q_dict, Qd_dict, R_dict, x_trj_d = make(a.T, h)
    if a.bounds is None:
        a.bounds = "abs" if hand else "rel"
    np.random.seed(a.seed)
    if a.method == "irs_lqr":
        params = amd.IrsLqrQuasistaticParameters()
        params.Q_dict, params.Qd_dict, params.R_dict = Q_dict, Qd_dict, R_dict
        params.x0, params.x_trj_d, params.u_trj_0, params.T = x0, x_trj_d, u_traj_0, a.T
        dim_u = q_dynamics.dim_u
        if a.bounds == "abs":       # run_planar_hand.py:138-139 (0.5 h); run_planar_hand_spin.py:142-143 (1.0 h)
            w = (1.0 if spin else 0.5) * h
            params.u_bounds_abs = np.array([-np.ones(dim_u) * w, np.ones(dim_u) * w])
        elif a.bounds == "rel":     # run_box_pivoting.py:119-120 (0.15 h); run_box_pushing.py:117-118 (0.4 h)
            w = 0.3 * h if hand else (0.4 * h if a.system == "box_pushing" else 0.15 * h)
            params.u_bounds_rel = np.array([-np.ones(dim_u) * w, np.ones(dim_u) * w])
        if spin:
            params.sampling = lambda u_initial, it: u_initial / (it ** 0.5)     # run_planar_hand_spin.py:146-151
            params.std_u_initial = np.ones(dim_u) * 0.1
        elif hand:
            params.sampling = lambda u_initial, it: u_initial / (it ** 0.8)     # run_planar_hand.py:142-146
            params.std_u_initial = np.ones(dim_u) * 0.3
        elif a.system == "box_pushing":
            params.sampling = lambda u_initial, it: u_initial ** (1.0 * it)     # run_box_pushing.py:120-124
            params.std_u_initial = np.ones(dim_u) * 0.3
        else:
            params.sampling = lambda u_initial, it: u_initial ** (0.5 * it)     # run_box_pivoting.py:122-126
            params.std_u_initial = np.ones(dim_u) * 0.1
        params.num_samples = a.N
        params.gradient_mode = a.gradient_mode or ("first_order" if hand else "zero_order_B")
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
