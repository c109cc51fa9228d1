"""Host mirror of the reference's cross-entropy-method baseline (irs_lqr/cem.py:7-216):
`CemParameters`, `CrossEntropyMethod(system, params)` with rollout / evaluate_cost / local_descent /
iterate and the attributes x_trj, u_trj, cost, std_trj, iter, x_trj_lst, u_trj_lst, cost_lst.

`local_descent` draws the candidates on the host exactly as the reference does
(`np.random.normal(u_trj, std_trj, (batch_size, T, m))`, cem.py:159-161 -- identical seeds give
identical candidates); the B rollouts + costs, the elite selection and the refit run on the GPU
(csrc/cem.hip).  Construction-time validation is IrsLqr's (same checks, same messages:
cem.py:77-106 duplicates irs_lqr.py:73-103).
"""
import time

import numpy as np

from . import device as dev
from .irs_lqr import IrsLqr


class CemParameters:
    """irs_lqr/cem.py:7-32 (same fields)."""

    def __init__(self):
        for name in ("Q", "Qd", "R", "x0", "xd_trj", "u_trj_initial", "n_elite", "batch_size", "elite_frac",
                     "initial_std"):
            setattr(self, name, None)


class CrossEntropyMethod:
    check_valid_system = IrsLqr.check_valid_system
    check_valid_params = IrsLqr.check_valid_params

    def __init__(self, system, params):
        self.system, self.params = system, params
        self.check_valid_system(system)
        self.check_valid_params(params, system)
        for name in ("Q", "Qd", "R", "x0", "xd_trj", "n_elite", "batch_size", "elite_frac", "initial_std"):
            setattr(self, name, getattr(params, name))
        self.u_trj = params.u_trj_initial
        self.T, self.dim_x, self.dim_u = self.u_trj.shape[0], system.dim_x, system.dim_u

        self._dm = system.dm()
        self._Q, self._R, self._x0, self._xd = (dev.to_dev(np.asarray(a, float))
                                                for a in (self.Q, self.R, self.x0, self.xd_trj))
        self.x_trj = self.rollout(self.x0, self.u_trj)
        self.cost = self.evaluate_cost(self.x_trj, self.u_trj)
        self.std_trj = np.tile(self.initial_std, (self.T, 1))
        self.x_trj_lst, self.u_trj_lst, self.cost_lst = [self.x_trj], [self.u_trj], [self.cost]
        self.start_time = time.time()
        self.iter = 1
        self.verbose = True

    def rollout(self, x0, u_trj):
        """cem.py:108-122."""
        x_trj, _ = self._dm.rollout_cost(dev.to_dev(np.asarray(x0, float)), dev.to_dev(np.asarray(u_trj, float)),
                                         self._Q, self._R, self._xd)
        return x_trj.cpu().numpy()

    def evaluate_cost(self, x_trj, u_trj):
        """cem.py:124-140 (terminal term with Q, like IrsLqr)."""
        return float(dev.evaluate_cost(dev.to_dev(np.asarray(x_trj, float)), dev.to_dev(np.asarray(u_trj, float)),
                                       self._Q, self._R, self._xd).item())

    def local_descent(self, x_trj, u_trj):
        """cem.py:151-184: sample, price, keep the elites, refit mean and std."""
        candidates = dev.to_dev(np.random.normal(u_trj, self.std_trj, (self.batch_size, self.T, self.dim_u)))
        self.cost_array = self._dm.cem_rollout_costs(candidates, self._x0, self._Q, self._R, self._xd)
        self.elite_idx, u_mean, u_std = self._dm.cem_refit(candidates, self.cost_array, self.n_elite)
        x_mean, _ = self._dm.rollout_cost(self._x0, u_mean, self._Q, self._R, self._xd)
        self.std_trj = u_std.cpu().numpy()
        return x_mean.cpu().numpy(), u_mean.cpu().numpy()

    def iterate(self, max_iterations):
        """cem.py:186-216: max_iterations + 1 descents, the last one logged but not adopted."""
        while True:
            x_new, u_new = self.local_descent(self.x_trj, self.u_trj)
            cost_new = self.evaluate_cost(x_new, u_new)
            if self.verbose:
                print("Iteration: {:02d}  ||  Current Cost: {:05f}  ||  Elapsed time: {:05f} ".format(
                    self.iter, cost_new, time.time() - self.start_time))
            for log, item in ((self.x_trj_lst, x_new), (self.u_trj_lst, u_new), (self.cost_lst, cost_new)):
                log.append(item)
            if self.iter > max_iterations:
                return self.x_trj, self.u_trj, self.cost
            self.cost, self.x_trj, self.u_trj = cost_new, x_new, u_new
            self.iter += 1
