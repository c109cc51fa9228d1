"""Host mirror of the reference's quasistatic cross-entropy-method baseline
(irs_lqr/cem_quasistatic.py:10-258): `CemQuasistaticParameters`,
`CrossEntropyMethodQuasistatic(q_dynamics, params)` with rollout / eval_cost / calc_Q_cost /
local_descent / iterate and the reference's attribute names (quasistatic_base.py lists them) plus
`n_elite, batch_size, initial_std, std_trj`.

`local_descent` draws the candidates on the host exactly as the reference does
(`np.random.normal(u_trj, std_trj, (batch_size, T, m))`, :188-189 -- identical seeds give identical
candidates); the B contact rollouts, their quasistatic costs, the elite selection and the refit run
on the GPU (csrc/cem.hip).  The reference's parameter class declares `xd_trj` but its solver reads
`params.x_trj_d` (:62, a latent AttributeError there): either attribute is accepted here.
"""
import numpy as np

from . import device as dev
from .quasistatic_base import QuasistaticOptimizerBase


class CemQuasistaticParameters:
    """irs_lqr/cem_quasistatic.py:10-37 (same fields)."""

    def __init__(self):
        for name in ("Q_dict", "Qd_dict", "R_dict", "x0", "xd_trj", "u_trj_0", "n_elite", "batch_size",
                     "initial_std", "T"):
            setattr(self, name, None)       # initial_std: (dim_u,) array of initial stds
        self.publish_every_iteration = True


class CrossEntropyMethodQuasistatic(QuasistaticOptimizerBase):
    def __init__(self, q_dynamics, params):
        goal = getattr(params, "x_trj_d", None)
        self._setup(q_dynamics, params, params.xd_trj if goal is None else goal)
        self.n_elite, self.batch_size, self.initial_std = params.n_elite, params.batch_size, params.initial_std
        self.std_trj = np.tile(self.initial_std, (self.T, 1))

    def local_descent(self, x_trj, u_trj):
        """cem_quasistatic.py:168-211: sample, price, keep the elites, refit mean and std."""
        candidates = dev.to_dev(np.random.normal(u_trj, self.std_trj, (self.batch_size, self.T, self.dim_u)))
        self.cost_array = self._dm.cem_rollout_costs_quasistatic(candidates, self._x0, self._Q, self._Qd, self._R,
                                                                 self._xd)
        self.elite_idx, u_mean, u_std = self._dm.cem_refit(candidates, self.cost_array, self.n_elite)
        x_mean, _ = self._dm.rollout_cost(self._x0, u_mean, self._Q, self._R, self._xd)
        self.std_trj = u_std.cpu().numpy()
        return x_mean.cpu().numpy(), u_mean.cpu().numpy()

    # outer loop: QuasistaticOptimizerBase.iterate
    def _start(self):
        return None

    def _descend(self, state):
        return self.local_descent(self.x_trj, self.u_trj) + (None,)
