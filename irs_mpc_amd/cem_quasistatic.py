"""Host mirror of the reference's quasistatic cross-entropy-method baseline
(irs_lqr/cem_quasistatic.py:10-258).

    CemQuasistaticParameters, CrossEntropyMethodQuasistatic(q_dynamics, params) with rollout /
    eval_cost / calc_Q_cost / local_descent / iterate and the attributes x_trj, u_trj, cost,
    std_trj, current_iter, x_trj_best, u_trj_best, cost_best, x_trj_list, u_trj_list,
    cost_all_list, cost_Qu_list, ...

`local_descent` draws the candidates on the host exactly as the reference does
(`np.random.normal(u_trj, std_trj, (batch_size, T, m))`, :188-189 -- identical seeds give identical
candidates); the B contact rollouts, their quasistatic costs, the elite selection and the refit run
on the GPU (csrc/cem.hip).  The reference's parameter class declares `xd_trj` but the solver reads
`params.x_trj_d` (:62, a latent AttributeError there): either attribute is accepted here.
"""
import time

import numpy as np

from . import device as dev
from .irs_lqr_quasistatic import quasistatic_eval_cost


class CemQuasistaticParameters:
    """irs_lqr/cem_quasistatic.py:10-37."""

    def __init__(self):
        self.Q_dict = None
        self.Qd_dict = None
        self.R_dict = None
        self.x0 = None
        self.xd_trj = None
        self.u_trj_0 = None
        self.n_elite = None
        self.batch_size = None
        self.initial_std = None  # dim u array of initial stds.
        self.T = None
        self.publish_every_iteration = True


class CrossEntropyMethodQuasistatic:
    def __init__(self, q_dynamics, params):
        self.q_dynamics = q_dynamics
        self.dim_x = q_dynamics.dim_x
        self.dim_u = q_dynamics.dim_u

        self.params = params

        self.T = params.T
        self.x0 = params.x0
        self.Q_dict = params.Q_dict
        self.Q = self.q_dynamics.get_Q_from_Q_dict(self.Q_dict)
        self.Qd_dict = params.Qd_dict
        self.Qd = self.q_dynamics.get_Q_from_Q_dict(self.Qd_dict)
        self.R_dict = params.R_dict
        self.R = self.q_dynamics.get_R_from_R_dict(self.R_dict)
        self.x_trj_d = getattr(params, "x_trj_d", None)
        if self.x_trj_d is None:
            self.x_trj_d = params.xd_trj
        self.u_trj_0 = params.u_trj_0
        self.indices_u_into_x = q_dynamics.get_u_indices_into_x()

        self.publish_every_iteration = params.publish_every_iteration

        self._dm = q_dynamics.dm()
        self._Q, self._Qd, self._R = (dev.to_dev(np.asarray(a, float)) for a in (self.Q, self.Qd, self.R))
        self._x0 = dev.to_dev(np.asarray(self.x0, float))
        self._xd = dev.to_dev(np.asarray(self.x_trj_d, float))

        self.x_trj = self.rollout(self.x0, self.u_trj_0)
        self.u_trj = self.u_trj_0  # T x m

        (cost_Qu, cost_Qu_final, cost_Qa, cost_Qa_final,
         cost_R) = self.eval_cost(self.x_trj, self.u_trj)
        self.cost = cost_Qu + cost_Qu_final + cost_Qa + cost_Qa_final + cost_R

        self.n_elite = params.n_elite
        self.batch_size = params.batch_size
        self.initial_std = params.initial_std
        self.std_trj = np.tile(self.initial_std, (self.T, 1))

        self.x_trj_best = None
        self.u_trj_best = None
        self.cost_best = np.inf

        # logging
        self.x_trj_list = [self.x_trj]
        self.u_trj_list = [self.u_trj]

        self.cost_all_list = [self.cost]
        self.cost_Qu_list = [cost_Qu]
        self.cost_Qu_final_list = [cost_Qu_final]
        self.cost_Qa_list = [cost_Qa]
        self.cost_Qa_final_list = [cost_Qa_final]
        self.cost_R_list = [cost_R]

        self.current_iter = 1
        self.start_time = time.time()
        self.verbose = True

    # irs_lqr/cem_quasistatic.py:101-108
    def rollout(self, x0, u_trj):
        T = u_trj.shape[0]
        assert T == self.T
        x_trj, _ = self._dm.rollout_cost(dev.to_dev(np.asarray(x0, float)), dev.to_dev(np.asarray(u_trj, float)),
                                         self._Q, self._R, self._xd)
        return x_trj.cpu().numpy()

    # irs_lqr/cem_quasistatic.py:110-165
    @staticmethod
    def calc_Q_cost(models_list, x_dict, xd_dict, Q_dict):
        cost = 0.
        for model in models_list:
            dx_i = x_dict[model] - xd_dict[model]
            cost += (dx_i * Q_dict[model] * dx_i).sum()
        return cost

    def eval_cost(self, x_trj, u_trj):
        T = u_trj.shape[0]
        assert T == self.T and x_trj.shape[0] == T + 1
        return quasistatic_eval_cost(self.q_dynamics, x_trj, u_trj, self.x_trj_d, self.Q_dict, self.Qd_dict, self.R)

    # irs_lqr/cem_quasistatic.py:168-211
    def local_descent(self, x_trj, u_trj):
        u_trj_mean = u_trj
        u_trj_candidates = np.random.normal(u_trj_mean, self.std_trj, (self.batch_size, self.T, self.dim_u))
        cand = dev.to_dev(u_trj_candidates)
        costs = self._dm.cem_rollout_costs_quasistatic(cand, self._x0, self._Q, self._Qd, self._R, self._xd)
        idx, u_new, std_new = self._dm.cem_refit(cand, costs, self.n_elite)
        self.cost_array = costs
        self.elite_idx = idx
        x_new, _ = self._dm.rollout_cost(self._x0, u_new, self._Q, self._R, self._xd)
        self.std_trj = std_new.cpu().numpy()
        return x_new.cpu().numpy(), u_new.cpu().numpy()

    # irs_lqr/cem_quasistatic.py:213-258
    def iterate(self, max_iterations):
        while True:
            if self.verbose:
                print('Iter {:02d},'.format(self.current_iter),
                      'cost: {:0.4f}.'.format(self.cost),
                      'time: {:0.2f}.'.format(time.time() - self.start_time))

            x_trj_new, u_trj_new = self.local_descent(self.x_trj, self.u_trj)
            (cost_Qu, cost_Qu_final, cost_Qa, cost_Qa_final,
             cost_R) = self.eval_cost(x_trj_new, u_trj_new)
            cost = cost_Qu + cost_Qu_final + cost_Qa + cost_Qa_final + cost_R
            self.x_trj_list.append(x_trj_new)
            self.u_trj_list.append(u_trj_new)
            self.cost_Qu_list.append(cost_Qu)
            self.cost_Qu_final_list.append(cost_Qu_final)
            self.cost_Qa_list.append(cost_Qa)
            self.cost_Qa_final_list.append(cost_Qa_final)
            self.cost_R_list.append(cost_R)
            self.cost_all_list.append(cost)

            if self.publish_every_iteration:
                self.q_dynamics.publish_trajectory(x_trj_new)

            if self.cost_best > cost:
                self.x_trj_best = x_trj_new
                self.u_trj_best = u_trj_new
                self.cost_best = cost

            if self.current_iter > max_iterations:
                break

            # Go over to next iteration.
            self.cost = cost
            self.x_trj = x_trj_new
            self.u_trj = u_trj_new
            self.current_iter += 1

        return self.x_trj, self.u_trj, self.cost
