"""Problem set-ups of the reference's analytic example scripts (same numbers, one place).

    pendulum      examples/pendulum/pendulum_{zero_order,first_order,exact,cem}.py
    quadrotor     examples/quadrotor/quadrotor_{zero_order,first_order,exact}.py
    bicycle[_hard] examples/bicycle/bicycle_{zero_order,first_order,exact,cem_*}[_hard].py
    three_cart    examples/three_cart/three_cart_{zero_order,cem}.py

Each entry returns (system, IrsLqrParameters, smoothing dict, cem dict, iterations).
"""
import numpy as np

import irs_mpc_amd as amd


def _params(Q, Qd, R, x0, xd_trj, u0, xb=None, ub=None):
    p = amd.IrsLqrParameters()
    p.Q, p.Qd, p.R = np.asarray(Q, float), np.asarray(Qd, float), np.asarray(R, float)
    p.x0, p.xd_trj, p.u_trj_initial = np.asarray(x0, float), xd_trj, u0
    p.xbound, p.ubound = xb, ub
    return p


def pendulum(T=200):
    # pendulum_zero_order.py:11-35
    sysd = amd.PendulumDynamics(0.05)
    p = _params(np.diag([1., 1.]), np.diag([20., 20.]), np.diag([1.]), [0, 0],
                np.tile(np.array([np.pi, 0.]), (T + 1, 1)), np.tile(np.array([0.1]), (T, 1)),
                [-np.array([1e4, 1e4]), np.array([1e4, 1e4])], np.array([[-1e4], [1e4]]))
    return sysd, p, dict(std_x=[1.0, 1.0], std_u=[1.0], N=1000, power=0.5), \
        dict(initial_std=np.array([1.0]), batch_size=1000, n_elite=10), 10


def quadrotor(T=200):
    # quadrotor_first_order.py:12-44
    sysd = amd.QuadrotorDynamics(0.05)
    xd = np.zeros((T + 1, 12))
    for i in range(T + 1):
        xd[i, :3] = [1.5 * np.cos(0.05 * i), 1.5 * np.sin(0.05 * i), 0.02 * i]
    big = np.array([1e5, 1e5, 1e5, 2.0 * np.pi, np.pi / 2, 2.0 * np.pi, 1e5, 1e5, 1e5, 1e5, 1e5, 1e5])
    p = _params(np.diag([10., 10, 10, 10, 10, 10, 0, 0, 0, 0, 0, 0]),
                10.0 * np.diag([10., 10, 10, 10, 10, 10, 1, 1, 1, 1, 1, 1]), np.eye(4), np.zeros(12), xd,
                np.tile(np.array([2.0, 2.0, 2.0, 2.0]), (T, 1)), [-big, big],
                np.array([-1e5 * np.ones(4), 1e5 * np.ones(4)]))
    return sysd, p, dict(std_x=0.1 * np.ones(12), std_u=0.1 * np.ones(4), N=1000, power=0.5), \
        dict(initial_std=0.5 * np.ones(4), batch_size=1000, n_elite=20), 3


def bicycle(T=100, hard=False):
    # bicycle_zero_order.py:11-36 ; bicycle_exact_hard.py:21 flips the goal
    sysd = amd.BicycleDynamics(0.1)
    goal = np.array([-3.0, -1.0, -np.pi / 2, 0, 0]) if hard else np.array([3.0, 1.0, np.pi / 2, 0, 0])
    p = _params(np.diag([5, 5, 3, 0.1, 0.1]), np.diag([50., 50, 30, 1, 1]), np.diag([1, 0.1]), np.zeros(5),
                np.tile(goal, (T + 1, 1)), np.tile(np.array([0.1, 0.0]), (T, 1)),
                [-np.array([1e4, 1e4, 1e4, 1e4, np.pi / 4]), np.array([1e4, 1e4, 1e4, 1e4, np.pi / 4])],
                np.array([[-1e4, -1e4], [1e4, 1e4]]))
    return sysd, p, dict(std_x=[2.0, 2.0, 1.0, 2.0, 0.01], std_u=[2.0, 1.0], N=10000, power=0.5), \
        dict(initial_std=np.array([1.0, 1.0]), batch_size=100, n_elite=10), 20


def bicycle_hard(T=100):
    return bicycle(T, hard=True)


def three_cart(T=100):
    # three_cart_zero_order.py:11-36
    sysd = amd.ThreeCartDynamics(0.05)
    p = _params(0.01 * np.diag([50., 50, 50, 20, 100, 20]), np.diag([50., 50, 50, 20, 100, 20]), 0.01 * np.eye(2),
                [0, 1, 2, 0, 0, 0], np.tile(np.array([2., 3, 4, 0, 0, 0]), (T + 1, 1)),
                np.tile(np.array([0.1, -0.1]), (T, 1)))
    return sysd, p, dict(std_x=4.0 * np.ones(6), std_u=0.5 * np.ones(2), N=1000, power=0.2), \
        dict(initial_std=0.5 * np.ones(2), batch_size=1000, n_elite=20), 20


PROBLEMS = {"pendulum": pendulum, "quadrotor": quadrotor, "bicycle": bicycle, "bicycle_hard": bicycle_hard,
            "three_cart": three_cart}
