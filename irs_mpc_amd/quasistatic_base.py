"""What the reference's two quasistatic optimisers have in common -- and duplicate verbatim
(irs_lqr/irs_lqr_quasistatic.py:44-194,347-390 and irs_lqr/cem_quasistatic.py:39-165,213-258):
problem set-up from per-model dicts, the five-term cost, the history lists, best-so-far tracking and
the outer loop.  Here it lives once; `IrsLqrQuasistatic` and `CrossEntropyMethodQuasistatic` only
supply `_start()` and `_descend(state)`.

Public attributes keep the reference's names: `x_trj, u_trj, cost, x_trj_best, u_trj_best, cost_best,
current_iter, x_trj_list, u_trj_list, cost_all_list, cost_Qu_list, cost_Qu_final_list, cost_Qa_list,
cost_Qa_final_list, cost_R_list, Q, Qd, R, Q_dict, Qd_dict, R_dict, x0, x_trj_d, u_trj_0, T,
indices_u_into_x`.
"""
import time

import numpy as np

from . import device as dev

# (reference attribute suffix, which models, running or final weight) of the four state-cost terms
_STATE_TERMS = (("Qu", "models_unactuated", False), ("Qu_final", "models_unactuated", True),
                ("Qa", "models_actuated", False), ("Qa_final", "models_actuated", True))
COST_TERMS = tuple(name for name, _, _ in _STATE_TERMS) + ("R",)


def quasistatic_eval_cost(q_dynamics, x_trj, u_trj, x_trj_d, Q_dict, Qd_dict, R):
    """(cost_Qu, cost_Qu_final, cost_Qa, cost_Qa_final, cost_R) of irs_lqr_quasistatic.py:153-194
    (= cem_quasistatic.py:124-165): per-model weighted state error, running rows with Q and the last
    row with Qd, and the input-RATE cost on u_t - u_{t-1} with u_{-1} = x_0[indices_u_into_x].
    Vectorised over time: O(T n) work on trajectories that are already on the host."""
    err = np.asarray(x_trj, float) - np.asarray(x_trj_d, float)
    terms = []
    for _, group, final in _STATE_TERMS:
        rows, weights = (err[-1:], Qd_dict) if final else (err[:-1], Q_dict)
        terms.append(sum(float(np.sum(rows[:, q_dynamics.position_indices[mdl]] ** 2 * np.asarray(weights[mdl], float)))
                         for mdl in getattr(q_dynamics, group)))
    held = np.asarray(x_trj, float)[0, q_dynamics.get_u_indices_into_x()]
    rate = np.diff(np.vstack([held[None], np.asarray(u_trj, float)]), axis=0)
    terms.append(float(np.einsum("ti,ij,tj->", rate, np.asarray(R, float), rate)))
    return tuple(terms)


class QuasistaticOptimizerBase:
    def _setup(self, q_dynamics, params, x_trj_d):
        """Problem data (reference names), their device copies, the initial rollout and the logs."""
        self.q_dynamics, self.params = q_dynamics, params
        self.dim_x, self.dim_u = q_dynamics.dim_x, q_dynamics.dim_u
        self.T, self.x0, self.u_trj_0, self.x_trj_d = params.T, params.x0, params.u_trj_0, x_trj_d
        self.Q_dict, self.Qd_dict, self.R_dict = params.Q_dict, params.Qd_dict, params.R_dict
        self.Q = q_dynamics.get_Q_from_Q_dict(self.Q_dict)
        self.Qd = q_dynamics.get_Q_from_Q_dict(self.Qd_dict)
        self.R = q_dynamics.get_R_from_R_dict(self.R_dict)
        self.indices_u_into_x = q_dynamics.get_u_indices_into_x()
        self.publish_every_iteration = params.publish_every_iteration

        self._dm = q_dynamics.dm()
        self._Q, self._Qd, self._R, self._x0, self._xd = (
            dev.to_dev(np.asarray(a, float)) for a in (self.Q, self.Qd, self.R, self.x0, self.x_trj_d))

        self.x_trj, self.u_trj = self.rollout(self.x0, self.u_trj_0), self.u_trj_0
        self.x_trj_list, self.u_trj_list, self.cost_all_list = [], [], []
        for name in COST_TERMS:
            setattr(self, "cost_%s_list" % name, [])
        self.x_trj_best = self.u_trj_best = None
        self.cost_best = np.inf
        self.cost = self._log(self.x_trj, self.u_trj, track_best=False)
        self.current_iter = 1
        self.start_time = time.time()
        self.verbose = True

    # ---- reference methods --------------------------------------------------------------
    def rollout(self, x0, u_trj):
        assert u_trj.shape[0] == self.T
        x_trj, _ = self._dm.rollout_cost(dev.to_dev(np.asarray(x0, float)), dev.to_dev(np.asarray(u_trj, float)),
                                         self._Q, self._R, self._xd)
        return x_trj.cpu().numpy()

    @staticmethod
    def calc_Q_cost(models_list, x_dict, xd_dict, Q_dict):
        return sum(float(((x_dict[mdl] - xd_dict[mdl]) ** 2 * Q_dict[mdl]).sum()) for mdl in models_list)

    def eval_cost(self, x_trj, u_trj):
        assert u_trj.shape[0] == self.T and x_trj.shape[0] == self.T + 1
        return quasistatic_eval_cost(self.q_dynamics, x_trj, u_trj, self.x_trj_d, self.Q_dict, self.Qd_dict, self.R)

    # ---- bookkeeping + outer loop ---------------------------------------------------------
    def _log(self, x_trj, u_trj, track_best=True):
        terms = self.eval_cost(x_trj, u_trj)
        total = sum(terms)
        self.x_trj_list.append(x_trj)
        self.u_trj_list.append(u_trj)
        self.cost_all_list.append(total)
        for name, value in zip(COST_TERMS, terms):
            getattr(self, "cost_%s_list" % name).append(value)
        if track_best and total < self.cost_best:
            self.x_trj_best, self.u_trj_best, self.cost_best = x_trj, u_trj, total
        return total

    def iterate(self, max_iterations):
        """max_iterations + 1 descents; the last one is logged (and may become the best) but not
        adopted -- the reference's loop shape (irs_lqr_quasistatic.py:347-390)."""
        state = self._start()
        while True:
            if self.verbose:
                print("Iter {:02d}, cost: {:0.4f}. time: {:0.2f}.".format(self.current_iter, self.cost,
                                                                        time.time() - self.start_time))
            x_new, u_new, state_new = self._descend(state)
            cost_new = self._log(x_new, u_new)
            if self.publish_every_iteration:
                self.q_dynamics.publish_trajectory(x_new)
            if self.current_iter > max_iterations:
                return self.x_trj, self.u_trj, self.cost
            self.cost, self.x_trj, self.u_trj, state = cost_new, x_new, u_new, state_new
            self.current_iter += 1
