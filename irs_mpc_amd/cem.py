"""Host mirror of the reference's cross-entropy-method baseline (irs_lqr/cem.py:7-216).

    CemParameters, CrossEntropyMethod(system, params) with rollout / evaluate_cost /
    local_descent / iterate and the attributes x_trj, u_trj, cost, std_trj, iter,
    x_trj_lst, u_trj_lst, cost_lst.

`local_descent` draws the candidates on the host exactly as the reference does
(`np.random.normal(u_trj, std_trj, (batch_size, T, m))`, cem.py:159-161 -- identical seeds
give identical candidates); the B rollouts + costs, the elite selection and the refit run
on the GPU (csrc/cem.hip).  (The reference declares `xd_trj` in CemParameters; its
quasistatic twin reads `x_trj_d` -- a latent bug there, not reproduced.)
"""
import time

import numpy as np

from . import device as dev


class CemParameters:
    """irs_lqr/cem.py:7-32."""

    def __init__(self):
        self.Q = None
        self.Qd = None
        self.R = None
        self.x0 = None
        self.xd_trj = None
        self.u_trj_initial = None
        self.n_elite = None
        self.batch_size = None
        self.elite_frac = None
        self.initial_std = None


class CrossEntropyMethod:
    def __init__(self, system, params):
        self.system = system
        self.params = params
        self.check_valid_system(self.system)
        self.check_valid_params(self.params, self.system)

        self.Q = params.Q
        self.Qd = params.Qd
        self.R = params.R
        self.x0 = params.x0
        self.xd_trj = params.xd_trj
        self.u_trj = params.u_trj_initial
        self.n_elite = params.n_elite
        self.batch_size = params.batch_size
        self.elite_frac = params.elite_frac
        self.initial_std = params.initial_std

        self.T = self.u_trj.shape[0]
        self.dim_x = self.system.dim_x
        self.dim_u = self.system.dim_u

        self._dm = system.dm()
        self._Q = dev.to_dev(np.asarray(self.Q, float))
        self._R = dev.to_dev(np.asarray(self.R, float))
        self._x0 = dev.to_dev(np.asarray(self.x0, float))
        self._xd = dev.to_dev(np.asarray(self.xd_trj, float))

        self.x_trj = self.rollout(self.x0, self.u_trj)
        self.cost = self.evaluate_cost(self.x_trj, self.u_trj)
        self.std_trj = np.tile(self.initial_std, (self.T, 1))

        self.x_trj_lst = [self.x_trj]
        self.u_trj_lst = [self.u_trj]
        self.cost_lst = [self.cost]
        self.start_time = time.time()
        self.iter = 1
        self.verbose = True

    # irs_lqr/cem.py:77-106
    def check_valid_system(self, system):
        if system.dim_x == 0:
            raise RuntimeError("System has zero states. Did you forget to set dim_x?")
        elif system.dim_u == 0:
            raise RuntimeError("System has zero inputs. Did you forget to set dim_u?")
        try:
            system.dynamics(np.zeros(system.dim_x), np.zeros(system.dim_u))
        except Exception:
            raise RuntimeError("Could not evaluate dynamics. Have you implemented it?")

    def check_valid_params(self, params, system):
        if params.Q.shape != (system.dim_x, system.dim_x):
            raise RuntimeError("Q matrix must be diagonal with dim_x x dim_x.")
        if params.Qd.shape != (system.dim_x, system.dim_x):
            raise RuntimeError("Qd matrix must be diagonal with dim_x x dim_x.")
        if params.R.shape != (system.dim_u, system.dim_u):
            raise RuntimeError("R matrix must be diagonal with dim_u x dim_u.")

    # irs_lqr/cem.py:108-140
    def rollout(self, x0, u_trj):
        x_trj, _ = self._dm.rollout_cost(dev.to_dev(np.asarray(x0, float)), dev.to_dev(np.asarray(u_trj, float)),
                                         self._Q, self._R, self._xd)
        return x_trj.cpu().numpy()

    def evaluate_cost(self, x_trj, u_trj):
        cost = dev.evaluate_cost(dev.to_dev(np.asarray(x_trj, float)), dev.to_dev(np.asarray(u_trj, float)),
                                 self._Q, self._R, self._xd)
        return float(cost.item())

    # irs_lqr/cem.py:151-184
    def local_descent(self, x_trj, u_trj):
        u_trj_mean = u_trj
        u_trj_candidates = np.random.normal(u_trj_mean, self.std_trj, (self.batch_size, self.T, self.dim_u))
        cand = dev.to_dev(u_trj_candidates)
        costs = self._dm.cem_rollout_costs(cand, self._x0, self._Q, self._R, self._xd)
        idx, u_new, std_new = self._dm.cem_refit(cand, costs, self.n_elite)
        self.cost_array = costs
        self.elite_idx = idx
        x_new, _ = self._dm.rollout_cost(self._x0, u_new, self._Q, self._R, self._xd)
        self.std_trj = std_new.cpu().numpy()
        return x_new.cpu().numpy(), u_new.cpu().numpy()

    # irs_lqr/cem.py:186-216
    def iterate(self, max_iterations):
        while True:
            x_trj_new, u_trj_new = self.local_descent(self.x_trj, self.u_trj)
            cost_new = self.evaluate_cost(x_trj_new, u_trj_new)
            if self.verbose:
                print("Iteration: {:02d} ".format(self.iter) + " || " +
                      "Current Cost: {0:05f} ".format(cost_new) + " || " +
                      "Elapsed time: {0:05f} ".format(time.time() - self.start_time))
            self.x_trj_lst.append(x_trj_new)
            self.u_trj_lst.append(u_trj_new)
            self.cost_lst.append(cost_new)
            if self.iter > max_iterations:
                break
            self.cost = cost_new
            self.x_trj = x_trj_new
            self.u_trj = u_trj_new
            self.iter += 1
        return self.x_trj, self.u_trj, self.cost
