"""Device-backed twins of the reference's analytic example plugins."""
import numpy as np
import torch

from . import device as dev
from ._lib import (SMOOTH_ZERO_ORDER_AB, SMOOTH_ZERO_ORDER_B,
                   MODEL_BICYCLE, MODEL_BOX_ON_BOX, MODEL_BOX_PIVOT, MODEL_BOX_PUSH, MODEL_PENDULUM,
                   MODEL_BOX_PIVOT_EXACT, MODEL_BOX_PUSH_EXACT, MODEL_PLANAR_HAND, MODEL_PLANAR_HAND_EXACT,
                   MODEL_QUADROTOR, MODEL_THREE_CART)
from .dynamical_system import DynamicalSystem


class PendulumDynamics(DynamicalSystem):
    """examples/pendulum/pendulum_dynamics.py:8-127: x=[angle, speed], u=[torque]."""
    device_model = MODEL_PENDULUM

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.dim_x = 2
        self.dim_u = 1


class QuadrotorDynamics(DynamicalSystem):
    """examples/quadrotor/quadrotor_dynamics.py:15-231: x=[xyz,rpy,xyz_d,rpy_d], u=rotors."""
    device_model = MODEL_QUADROTOR

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.dim_x = 12
        self.dim_u = 4
        # quadrotor_dynamics.py:26-37
        self.m = 0.775
        self.L = 0.15
        self.g = 9.81
        self.Ixx, self.Iyy, self.Izz = 0.0015, 0.0025, 0.0035
        self.kF = 1.0
        self.kM = 0.0245

    def device_params(self):
        return [self.h, self.m, self.L, self.g, self.Ixx, self.Iyy, self.Izz, self.kF, self.kM]


class BicycleDynamics(DynamicalSystem):
    """examples/bicycle/bicycle_dynamics.py:8-132: x=[x,y,heading,speed,steer], u=[accel,steer rate]."""
    device_model = MODEL_BICYCLE

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.dim_x = 5
        self.dim_u = 2


class ThreeCartDynamics(DynamicalSystem):
    """examples/three_cart/three_cart_dynamics.py:8-107: x=[q1,q2,q3,v1,v2,v3], u=[u1,u3].
    Follows the scalar `dynamics` (the reference's `dynamics_batch` resolves penetration
    differently, :175-188).  The reference offers no Jacobian for it; here `jacobian_xu`
    returns the active branch's derivative."""
    device_model = MODEL_THREE_CART

    def __init__(self, dt):
        super().__init__()
        self.h = dt
        self.dim_x = 6
        self.dim_u = 2
        self.d = 0.2

    def device_params(self):
        return [self.h, self.d]


class QuasistaticDeviceDynamics(DynamicalSystem):
    """What `QuasistaticDynamics` (irs_lqr/quasistatic_dynamics.py:15-164) offers besides the step:
    the bookkeeping it derives from the plant (:22-28), keyed by model NAME here (the reference keys
    by ModelInstanceIndex).  Subclasses set `models_unactuated`, `models_actuated`,
    `position_indices` and a device contact functor.  `jacobian_xu*` return the simulator's
    `[Dq_nextDq | Dq_nextDqa_cmd]` (:184-191): the derivative of the step QP through its active
    constraints, contact geometry held fixed.  The sample-pass modes (`ZERO_ORDER_B`, `FIRST_ORDER`)
    return the decoupled (A,B) of irs_lqr_quasistatic.py:275-284."""

    def calc_AB_exact(self, x_nominal, u_nominal):
        """quasistatic_dynamics.py:189-191."""
        return self.jacobian_xu(x_nominal, u_nominal)

    # ---- the randomised-smoothing estimators (quasistatic_dynamics.py:193-300) ---------------------------
    # Same names, arguments, draws (np.random.normal from the global generator, in the reference's order:
    # per nominal point, dx before du) and return layout `[A | B]`, (n, n+m) -- the UNDECOUPLED pair, as in
    # the reference, where IrsLqrQuasistatic.decouple_AB_matrices comes afterwards.  Underneath, each is
    # the launch(es) `calc_AB_batch` makes over all nominal points at once.
    ZERO_ORDER_AB_STD_X = 1e-3      # calc_AB_zero_order's defaults (quasistatic_dynamics.py:270-271)
    ZERO_ORDER_AB_DAMP = 1e-2
    AB_MODES = ("first_order", "zero_order_B", "zero_order_AB", "exact")

    def calc_AB_first_order(self, x_nominal, u_nominal, n_samples, std_u):
        """quasistatic_dynamics.py:193-208: the mean over `n_samples` u-perturbed steps of the simulator's
        `[Dq_nextDq | Dq_nextDqa_cmd]`."""
        return self.calc_AB_batch(np.asarray(x_nominal, float)[None], np.asarray(u_nominal, float)[None],
                                  n_samples, std_u, "first_order")[0]

    def calc_B_zero_order(self, x_nominal, u_nominal, n_samples, std_u):
        """quasistatic_dynamics.py:242-266: B by least squares on u-perturbed steps, A = `Dq_nextDq` at the
        nominal point."""
        return self.calc_AB_batch(np.asarray(x_nominal, float)[None], np.asarray(u_nominal, float)[None],
                                  n_samples, std_u, "zero_order_B")[0]

    def calc_AB_zero_order(self, x_nominal, u_nominal, n_samples, std_u, std_x=1e-3, damp=1e-2):
        """quasistatic_dynamics.py:268-300: A and B by least squares on (x, u)-perturbed steps with `damp`
        identity rows appended."""
        return self.calc_AB_batch(np.asarray(x_nominal, float)[None], np.asarray(u_nominal, float)[None],
                                  n_samples, std_u, "zero_order_AB", std_x=std_x, damp=damp)[0]

    def calc_AB_batch(self, x_nominals, u_nominals, n_samples, std_u, mode, std_x=1e-3, damp=1e-2):
        """quasistatic_dynamics.py:210-240: `ABhat_list` (k, n, n+m) for k nominal points; `mode` one of
        "first_order", "zero_order_B", "zero_order_AB", "exact" (anything else: the reference's
        RuntimeError, :238)."""
        x = dev.to_dev(np.atleast_2d(np.asarray(x_nominals, float)))
        u = dev.to_dev(np.atleast_2d(np.asarray(u_nominals, float)))
        At, Bt, _, info = self.calc_AB_batch_dev(x, u, n_samples, std_u, mode, std_x=std_x, damp=damp)
        if bool((info != 0).any().item()):
            raise ValueError("randomized-smoothing least squares is rank deficient or met a non-finite sample")
        return torch.cat([At, Bt], dim=2).cpu().numpy()

    def calc_AB_batch_dev(self, x, u, n_samples, std_u, mode, std_x=1e-3, damp=1e-2, seed=None, it=1):
        """`calc_AB_batch` on device tensors: x (k,n), u (k,m) f64 -> (At (k,n,n), Bt (k,n,m), ct (k,n), info (k))
        with c = f(x,u) - A x - B u.  One sample-pass launch over all k x n_samples perturbed steps (zero-order
        modes: + the solve on the sums; "zero_order_B": + one launch of k f64 lanes for A; "first_order": the
        f64 Jacobian lanes, one per sample).  `seed` None: the perturbations are the reference's host draws;
        an int: drawn on the device (Philox, iteration counter `it`).  With torch.distributed initialised every
        rank draws the same samples, keeps its shard and the statistics are all-reduced once."""
        from . import distributed as dist_util
        if mode not in self.AB_MODES:
            raise RuntimeError(f"AB mode {mode} is not supported.")                 # quasistatic_dynamics.py:238
        dm, n, m, k = self.dm(), self.dim_x, self.dim_u, x.shape[0]
        N = int(n_samples)
        std_u = np.broadcast_to(np.asarray(std_u, float), (m,))
        zero_info = torch.zeros(k, dtype=torch.int32, device=x.device)
        if mode == "exact":
            return (*dm.exact_linearize(x, u), zero_info)
        rank, world = dist_util.rank_world()
        lo, hi = dist_util.shard_range(N, rank, world)

        def c_of(At, Bt):
            f = dm.dynamics_batch(x, u)
            return (f - torch.einsum("tij,tj->ti", At, x) - torch.einsum("tij,tj->ti", Bt, u)).contiguous()

        if mode == "first_order":
            if seed is None:
                du = np.stack([np.random.normal(0, std_u, size=[N, m]) for _ in range(k)])
                du = dev.to_dev(np.ascontiguousarray(du[:, lo:hi]))
            else:
                du = dm.rng_samples(k, hi - lo, np.zeros(n), std_u, int(seed), int(it), sample_offset=lo)[1].to(dev.F64)
            sums = torch.zeros((k, n * (n + m)), dtype=dev.F64, device=x.device)
            step = max(1, 200000 // max(hi - lo, 1))          # <= 2e5 Jacobian lanes (123 MB of f64) per launch
            for a in range(0, k, step):
                b = min(k, a + step)
                X = x[a:b, None, :].expand(b - a, hi - lo, n).reshape(-1, n).contiguous()
                U = (u[a:b, None, :] + du[a:b]).reshape(-1, m).contiguous()
                sums[a:b] = dm.jacobian_xu_batch(X, U).reshape(b - a, hi - lo, -1).sum(1)
            dist_util.all_reduce_sums(sums)
            AB = (sums / float(N)).reshape(k, n, n + m)
            At, Bt = AB[:, :, :n].contiguous(), AB[:, :, n:].contiguous()
            return At, Bt, c_of(At, Bt), zero_info

        if mode == "zero_order_AB":
            MODE = SMOOTH_ZERO_ORDER_AB
            sx = np.broadcast_to(np.asarray(std_x, float), (n,))
            if seed is None:
                dxs, dus = [], []
                for _ in range(k):                                                   # :282-283: dx first
                    dxs.append(np.random.normal(0, sx, size=[N, n]))
                    dus.append(np.random.normal(0, std_u, size=[N, m]))
                dxd = dev.to_dev(np.ascontiguousarray(np.stack(dxs)[:, lo:hi], np.float32), dev.F32)
                dud = dev.to_dev(np.ascontiguousarray(np.stack(dus)[:, lo:hi], np.float32), dev.F32)
                sums = dm.smooth_accumulate(MODE, x, u, dxd, dud)
            else:
                sums = dm.smooth_accumulate_rng(MODE, x, u, hi - lo, sx, std_u, int(seed), int(it), sample_offset=lo)
            dist_util.all_reduce_sums(sums)
            # the `damp` identity rows add damp^2 to the Gram diagonal and nothing to the cross term
            d = n + m
            diag = torch.as_tensor([i * d - i * (i - 1) // 2 for i in range(d)], device=sums.device)
            sums[:, diag] += float(damp) ** 2
            ws = dm._workspace(MODE, k, hi - lo, x.device)
            return dm.smooth_finalize(MODE, N, x, u, sums, workspace=ws)

        # "zero_order_B": statistics [upper Gram of du | du (f - xb)' | sum du], xb = the f32-rounded nominal state
        MODE = SMOOTH_ZERO_ORDER_B
        if seed is None:
            du = np.stack([np.random.normal(0, std_u, size=[N, m]) for _ in range(k)])
            du = dev.to_dev(np.ascontiguousarray(du[:, lo:hi], np.float32), dev.F32)
            sums = dm.smooth_accumulate(MODE, x, u, None, du)
        else:
            sums = dm.smooth_accumulate_rng(MODE, x, u, hi - lo, None, std_u, int(seed), int(it), sample_offset=lo)
        dist_util.all_reduce_sums(sums)
        return self.zero_order_B_from_sums_dev(x, u, sums)

    def zero_order_B_from_sums_dev(self, x, u, sums):
        """calc_B_zero_order from the (all-reduced) ZERO_ORDER_B statistics of a contact model
        (include/irs_hip.h): B = the least-squares fit, A = the step's derivative at the nominal point."""
        dm, n, m, k = self.dm(), self.dim_x, self.dim_u, x.shape[0]
        Ae, Be, ce = dm.exact_linearize(x, u)
        f = ce + torch.einsum("tij,tj->ti", Ae, x) + torch.einsum("tij,tj->ti", Be, u)
        iu = torch.triu_indices(m, m, device=sums.device)
        ng = iu.shape[1]
        G = torch.zeros((k, m, m), dtype=sums.dtype, device=sums.device)
        G[:, iu[0], iu[1]] = sums[:, :ng]
        G = G + torch.triu(G, 1).transpose(1, 2)
        H = sums[:, ng:ng + m * n].reshape(k, m, n)
        sz = sums[:, ng + m * n:ng + m * n + m]
        xb = x.to(torch.float32).to(sums.dtype)
        H = H - sz[:, :, None] * (f - xb)[:, None, :]
        Bt, info = torch.linalg.solve_ex(G, H)
        Bt = Bt.transpose(1, 2).contiguous()
        bad = (info != 0) | ~torch.isfinite(Bt).reshape(k, -1).all(1)
        ct = (f - torch.einsum("tij,tj->ti", Ae, x) - torch.einsum("tij,tj->ti", Bt, u)).contiguous()
        return Ae, Bt, ct, bad.to(torch.int32)

    def _finish_bookkeeping(self):
        self.models_all = self.models_unactuated + self.models_actuated
        self.velocity_indices = self.position_indices

    # ---- quasistatic_dynamics.py:57-130: vector <-> per-model dict helpers --------------
    def get_u_indices_into_x(self):
        return np.concatenate([self.position_indices[m] for m in self.models_actuated])

    def get_q_a_cmd_dict_from_u(self, u):
        out, i = {}, 0
        for m in self.models_actuated:
            k = len(self.position_indices[m])
            out[m] = u[i:i + k]
            i += k
        return out

    def get_q_dict_from_x(self, x):
        return {m: x[idx] for m, idx in self.position_indices.items()}

    def get_x_from_q_dict(self, q_dict):
        x = np.zeros(self.dim_x)
        for m, idx in self.position_indices.items():
            x[idx] = q_dict[m]
        return x

    def get_u_from_q_cmd_dict(self, q_cmd_dict):
        return np.concatenate([np.asarray(q_cmd_dict[m], float) for m in self.models_actuated])

    def get_Q_from_Q_dict(self, Q_dict):
        Q = np.eye(self.dim_x)
        for m, idx in self.velocity_indices.items():
            Q[idx, idx] = Q_dict[m]
        return Q

    def get_R_from_R_dict(self, R_dict):
        R = np.eye(self.dim_u)
        i = 0
        for m in self.models_actuated:
            k = len(self.position_indices[m])
            R[i:i + k, i:i + k] = np.diag(R_dict[m])
            i += k
        return R

    def publish_trajectory(self, x_traj):
        """quasistatic_dynamics.py:132-135 animates in meshcat; there is no visualiser here."""



class PlanarHandDynamics(QuasistaticDeviceDynamics):
    """Device twin of `QuasistaticDynamics` (irs_lqr/quasistatic_dynamics.py:15-164) for the
    planar hand of examples/planar_hand (planar_hand_setup.py:8-27): a disc cradled by two
    2-link arms, x = [xo, ql1, qr1, yo, ql2, qr2, th] (the reference's order), u = commanded joint
    angles [ql1, ql2, qr1, qr2].  Steps the
    Anitescu convex quasi-dynamic QP on the device (csrc/contact_models.hpp); the reference steps
    the external quasistatic_simulator, so parity for this model is UNPINNED."""
    device_model = MODEL_PLANAR_HAND

    def __init__(self, h, mass=1.0, mu=0.5, pgs_iters=50, contact_solver="exact"):
        """contact_solver: "exact" (default) = the step QP solved exactly by the dual active-set method -- what
        the reference's simulator does (it hands every step QP to Gurobi, quasistatic_dynamics.py:146-164);
        "pgs" = `pgs_iters` over-relaxed projected sweeps + an active-set polish (opt-in: 0.7x the time,
        approximate on the few per cent of samples where many contacts load the disc)."""
        super().__init__()
        if contact_solver not in ("pgs", "exact"):
            raise ValueError("contact_solver must be 'pgs' or 'exact'")
        if contact_solver == "exact":
            self.device_model = MODEL_PLANAR_HAND_EXACT
        self.contact_solver = contact_solver
        self.h = h
        self.dim_x = 7
        self.dim_u = 4
        self.g = 10.0            # planar_hand_setup.py:23
        self.mass = mass
        self.R = 0.25            # planar_hand_setup.py:8
        self.mu = mu
        self.kp = (50.0, 25.0)   # planar_hand_setup.py:12
        self.l1, self.l2 = 0.3, 0.2
        self.r_link = 0.05
        self.base_x = 0.1
        self.pgs_iters = pgs_iters

        # the bookkeeping QuasistaticDynamics derives from the plant (quasistatic_dynamics.py:22-28),
        # keyed by model NAME here (the reference keys by ModelInstanceIndex)
        self.models_unactuated = ["sphere"]
        self.models_actuated = ["arm_left", "arm_right"]
        # the reference's state order (planar_hand_analysis.py:61-67): x = [xo, ql1, qr1, yo, ql2, qr2, th]
        self.position_indices = {"sphere": np.array([0, 3, 6]), "arm_left": np.array([1, 4]),
                                 "arm_right": np.array([2, 5])}
        self._finish_bookkeeping()

    def device_params(self):
        return [self.h, self.g, self.mass, self.R, self.mu, self.kp[0], self.kp[1], self.l1, self.l2,
                self.r_link, self.base_x, float(self.pgs_iters)]


class BoxPivotingDynamics(QuasistaticDeviceDynamics):
    """Device twin of `QuasistaticDynamics` for examples/box_pivoting (box_pivoting_setup.py:6-19,
    run_box_pivoting.py:20-75): a 1 m square box on the ground, pivoted by a position-controlled disc.
    x = [x_h, x_b, y_h, y_b, th_b] (the reference's order, analysis/box_pivoting_analysis.py:53-64),
    u = commanded hand position.  Same contact scheme as the planar hand; parity UNPINNED."""
    device_model = MODEL_BOX_PIVOT

    EXACT_MODEL = MODEL_BOX_PIVOT_EXACT

    def __init__(self, h, mass=1.0, mu=0.5, pgs_iters=50, contact_solver="exact"):
        """contact_solver: as for PlanarHandDynamics ("exact" = the reference's semantics, default)."""
        super().__init__()
        if contact_solver not in ("pgs", "exact"):
            raise ValueError("contact_solver must be 'pgs' or 'exact'")
        if contact_solver == "exact":
            self.device_model = type(self).EXACT_MODEL
        self.contact_solver = contact_solver
        self.h = h
        self.dim_x = 5
        self.dim_u = 2
        self.g = 9.81            # box_pivoting_setup.py:19
        self.mass = mass
        self.half = 0.5          # box_1m_rotation.sdf (box_pivoting_setup.py:6)
        self.mu = mu
        self.kp = 50000.0        # box_pivoting_setup.py:10
        self.r_hand = 0.1        # analysis/box_pivoting_analysis.py:61
        self.pgs_iters = pgs_iters
        self.models_unactuated = ["box"]
        self.models_actuated = ["hand"]
        self.position_indices = {"box": np.array([1, 3, 4]), "hand": np.array([0, 2])}
        self._finish_bookkeeping()

    def device_params(self):
        return [self.h, self.g, self.mass, self.half, self.mu, self.kp, self.r_hand, float(self.pgs_iters)]


class BoxOnBoxDynamics(QuasistaticDeviceDynamics):
    """The reference's 1-D statement of the quasi-dynamic step (examples/box_pushing/analysis/
    box_on_box.py:11-20: m = 1, k = 100, h = 0.1): x = [x_a, x_u], u = commanded x_a.  Runs the same
    device contact-QP code as the planar functors and is checked against the closed form there."""
    device_model = MODEL_BOX_ON_BOX

    def __init__(self, h=0.1, m=1.0, k=100.0, pgs_iters=50):
        super().__init__()
        self.h, self.m, self.k, self.pgs_iters = h, m, k, pgs_iters
        self.dim_x, self.dim_u = 2, 1
        self.models_unactuated, self.models_actuated = ["box"], ["pusher"]
        self.position_indices = {"pusher": np.array([0]), "box": np.array([1])}
        self._finish_bookkeeping()

    def device_params(self):
        return [self.h, self.m, self.k, float(self.pgs_iters)]


class BoxPushingDynamics(BoxPivotingDynamics):
    """Device twin of `QuasistaticDynamics` for examples/box_pushing (box_pushing_setup.py:6-19,
    run_box_pushing.py:20-75): the box and disc of box_pivoting seen from above -- no gravity, no
    ground, Kp = 500.  x = [x_h, x_b, y_h, y_b, th_b], u = commanded hand position.  PINNED: mass 5,
    inertia 1/6, half side 0.4995 and r_hand 0.1 are identified from the simulator data the reference
    ships (examples/box_pushing/analysis/{xu,dxdu}_quasistatic.npy); the step reproduces the recorded
    trajectory to 3e-8 and `jacobian_xu` the recorded Jacobians to 5e-7."""
    device_model = MODEL_BOX_PUSH
    EXACT_MODEL = MODEL_BOX_PUSH_EXACT

    def __init__(self, h=0.1, mass=5.0, inertia=1.0 / 6.0, mu=0.5, pgs_iters=50, contact_solver="exact"):
        super().__init__(h, mass, mu, pgs_iters, contact_solver)
        self.g = 0.0             # box_pushing_setup.py:18
        self.inertia = inertia
        self.kp = 500.0          # box_pushing_setup.py:10
        self.half = 0.4995       # lever arm of the contact point in the simulator's sticking Jacobians
        self.r_hand = 0.1        # touching distance 0.5995 in the recorded push

    def device_params(self):
        return [self.h, self.mass, self.inertia, self.half, self.mu, self.kp, self.r_hand, float(self.pgs_iters)]
