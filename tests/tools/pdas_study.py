"""CPU study of the active-set iterations of the bounded quasistatic descent (oracle twin, oracle/irs_oracle.py
ctrlbox_solve) on inputs dumped from the device loop (tools/descent_loop_profile.py --dump=K):
    python tests/tools/pdas_study.py gpurun_out/descent_inputs_box_pivoting_it2.npz [strategy ...]
Prints, per strategy, the tails whose primal-dual phase did not converge and the totals (iterations, backward time
steps swept, forward time steps) -- the quantities the device kernel's time is made of."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import irs_oracle as orc  # noqa: E402


WINDOW = int(os.environ.get("PDAS_WINDOW", "0"))
FROZEN_TOTAL = []
DEVICE_MU = os.environ.get("PDAS_DEVICE_MU", "1") == "1"


def solve(prob, s_start, t0, lo, hi, u, act, W, valid_from, strategy, pdas_iter=10, max_iter=2000, tol=1e-10):
    """ctrlbox_solve with a choice of phase-1 strategy; returns additionally the forward steps rolled."""
    T, m = prob["B"].shape[0], prob["m"]
    A, B, c = prob["A"], prob["B"], prob["c"]
    s = np.zeros((T + 1, A.shape[1]))
    mu = np.zeros((T, m))
    cnt = dict(swept=0, fwd=0)
    FROZEN_TOTAL.append(cnt)

    def backward(t_hi):
        if t_hi >= t0:
            orc.ctrlbox_backward(prob, act, lo, hi, t_hi, t0, W)
            cnt["swept"] += t_hi - t0 + 1

    def policy_rollout(us, clipped=False):
        s[t0] = s_start
        for t in range(t0, T):
            us[t] = W["K"][t] @ s[t] + W["k"][t]
            uc = np.clip(us[t], lo[t], hi[t]) if clipped else us[t]
            # (the device's multiplier rows are Y s~ = H K s + ... : the policy's own control, not the clipped one)
            mu[t] = W["H"][t] @ (us[t] if DEVICE_MU else uc) + W["G"][t] @ s[t] + W["g"][t]
            s[t + 1] = A[t] @ s[t] + B[t] @ uc + c[t]
        cnt["fwd"] += T - t0

    t_dirty = T - 1 if valid_from >= T else (valid_from - 1 if valid_from > t0 else t0 - 1)
    # composable: "wr10_60+clip3" = the damped rule from round 11 on, the first three rounds of a tail clipped
    clipk = 0
    if "+" in strategy:
        strategy, extra = strategy.split("+", 1)
        if extra.startswith("clip"):
            clipk = int(extra[4:])
    elif strategy.startswith("clip"):
        clipk = int(strategy[4:])
    n1 = pdas_iter
    if strategy.startswith("more"):
        n1 = int(strategy[4:])
    wr_after = None
    if strategy.startswith("wr"):           # wrK_N: all adds + only the worst release from iteration K+1 on, N iterations
        wr_after, n1 = (int(v) for v in strategy[2:].split("_"))
    if strategy == "frozen":
        return solve_frozen(prob, s_start, t0, lo, hi, u, act, W, valid_from, clipk > 0, backward, policy_rollout, cnt, s, mu,
                            tol)
    seen = {}
    it = 0
    mode_add_only = False
    while it < n1:
        it += 1
        backward(t_dirty)
        # clipK: the first K rounds of a tail roll the policy out with the controls CLIPPED to their boxes
        policy_rollout(u, clipped=it <= clipk)
        a, uu, mm = act[t0:], u[t0:], mu[t0:]
        new = a.copy()
        add_lo = (a == 0) & (uu < lo[t0:] - tol)
        add_hi = (a == 0) & (uu > hi[t0:] + tol)
        rel = ((a < 0) & (mm < -tol)) | ((a > 0) & (mm > tol))
        if strategy == "alt" and it > 4:
            # primal feasibility first: release only when nothing is left to add
            if add_lo.any() or add_hi.any():
                rel = np.zeros_like(rel)
        if (strategy == "worstrel" and it > 4) or (wr_after is not None and it > wr_after):
            # all adds, but only the single worst release
            if rel.any():
                viol = np.where(rel, np.abs(mm), 0.0)
                r, j = np.unravel_index(np.argmax(viol), viol.shape)
                rel = np.zeros_like(rel)
                rel[r, j] = True
        if strategy.startswith("tabu"):     # tabuF_N: an index that changed F times is released only as the single worst
            F, n1 = (int(v) for v in strategy[4:].split("_"))
            if "flips" not in seen:
                seen["flips"] = np.zeros(act.shape, dtype=int)
            fl = seen["flips"][t0:]
            hot = rel & (fl >= F)
            if hot.any():
                viol = np.where(hot, np.abs(mm), 0.0)
                r, j = np.unravel_index(np.argmax(viol), viol.shape)
                keep = np.zeros_like(rel)
                keep[r, j] = True
                rel = (rel & ~hot) | keep
        if strategy.startswith("frac"):     # fracP_K_N: from iteration K+1 on release those within P % of the worst multiplier
            P, K, n1 = (int(v) for v in strategy[4:].split("_"))
            if it > K and rel.any():
                viol = np.where(rel, np.abs(mm), 0.0)
                rel = rel & (viol >= viol.max() * P / 100.0)
        if strategy.startswith("top"):      # topR_K_N: from iteration K+1 on release the R worst
            Rn, K, n1 = (int(v) for v in strategy[3:].split("_"))
            if it > K and rel.sum() > Rn:
                viol = np.where(rel, np.abs(mm), 0.0)
                thr = np.sort(viol.ravel())[-Rn]
                rel = rel & (viol >= thr)
        if strategy.startswith("early"):    # earlyK_N: from iteration K+1 on release only the EARLIEST time step's wrong-signed ones
            K, n1 = (int(v) for v in strategy[5:].split("_"))
            if it > K and rel.any():
                r0 = np.nonzero(rel.any(axis=1))[0].min()
                keep = np.zeros_like(rel)
                keep[r0] = rel[r0]
                rel = keep
        if strategy.startswith("late"):     # lateK_N: ... only the LATEST time step's
            K, n1 = (int(v) for v in strategy[4:].split("_"))
            if it > K and rel.any():
                r0 = np.nonzero(rel.any(axis=1))[0].max()
                keep = np.zeros_like(rel)
                keep[r0] = rel[r0]
                rel = keep
        if strategy == "cyc":
            key = a.tobytes()
            if key in seen:
                mode_add_only = True
            seen[key] = it
            if mode_add_only and (add_lo.any() or add_hi.any()):
                rel = np.zeros_like(rel)
        if WINDOW > 0:
            # only the changes within WINDOW steps of the earliest one (shorter re-sweeps, possibly more rounds)
            anyc = (add_lo | add_hi | rel).any(axis=1)
            if anyc.any():
                first = int(np.nonzero(anyc)[0].min())
                keep = np.zeros(a.shape[0], dtype=bool)
                keep[first:first + WINDOW] = True
                add_lo = add_lo & keep[:, None]
                add_hi = add_hi & keep[:, None]
                rel = rel & keep[:, None]
        new[add_lo] = -1
        new[add_hi] = 1
        new[rel] = 0
        changed = np.nonzero((new != a).any(axis=1))[0]
        if changed.size == 0:
            return s, u, mu, (it, 0, cnt["swept"], cnt["fwd"]), t0
        if "flips" in seen:
            seen["flips"][t0:] += (new != a)
        act[t0:] = new
        t_dirty = t0 + int(changed.max())
    # phase 2 as the oracle's
    u[t0:] = np.clip(u[t0:], lo[t0:], hi[t0:])
    new = np.where(u[t0:] <= lo[t0:], -1, np.where(u[t0:] >= hi[t0:], 1, 0))
    changed = np.nonzero((new != act[t0:]).any(axis=1))[0]
    act[t0:] = new
    t_dirty = max(t_dirty, t0 + int(changed.max())) if changed.size else t_dirty
    us = np.zeros_like(u)
    for it2 in range(1, max_iter + 1):
        backward(t_dirty)
        t_dirty = t0 - 1
        policy_rollout(us)
        d = us[t0:] - u[t0:]
        free = act[t0:] == 0
        with np.errstate(divide="ignore", invalid="ignore"):
            room = np.where(d > 0, (hi[t0:] - u[t0:]) / d, np.where(d < 0, (lo[t0:] - u[t0:]) / d, np.inf))
        room = np.where(free, room, np.inf)
        alpha = min(1.0, room.min())
        if alpha < 1.0:
            r, j = np.unravel_index(np.argmin(room), room.shape)
            u[t0:] += alpha * d
            act[t0 + r, j] = 1 if d[r, j] > 0 else -1
            u[t0 + r, j] = hi[t0 + r, j] if d[r, j] > 0 else lo[t0 + r, j]
            t_dirty = t0 + r
            continue
        u[t0:] = us[t0:]
        viol = np.where(act[t0:] < 0, -mu[t0:], np.where(act[t0:] > 0, mu[t0:], 0.0))
        if viol.max() <= tol:
            return s, u, mu, (it, it2, cnt["swept"], cnt["fwd"]), t0
        r, j = np.unravel_index(np.argmax(viol), viol.shape)
        act[t0 + r, j] = 0
        t_dirty = t0 + r
    return s, u, mu, (it, -max_iter, cnt["swept"], cnt["fwd"]), t0


def solve_frozen(prob, s_start, t0, lo, hi, u, act, W, valid_from, clip, backward, policy_rollout, cnt, s, mu, tol):
    """Outer rounds = exact sweeps; inner rounds keep the cost-to-go FROZEN and recompute the gains of the changed steps
    only (each from its stored P_{t+1}), then roll out again.  cnt["frozen"] counts the single-step updates."""
    T = prob["B"].shape[0]
    cnt["frozen"] = 0
    t_dirty = T - 1 if valid_from >= T else (valid_from - 1 if valid_from > t0 else t0 - 1)

    def update():
        a, uu, mm = act[t0:], u[t0:], mu[t0:]
        new = a.copy()
        new[(a == 0) & (uu < lo[t0:] - tol)] = -1
        new[(a == 0) & (uu > hi[t0:] + tol)] = 1
        new[((a < 0) & (mm < -tol)) | ((a > 0) & (mm > tol))] = 0
        changed = np.nonzero((new != a).any(axis=1))[0]
        act[t0:] = new
        return changed

    outer = 0
    while outer < 60:
        outer += 1
        backward(t_dirty)                     # exact
        policy_rollout(u, clipped=clip)
        changed = update()
        if changed.size == 0:
            return s, u, mu, (outer, 0, cnt["swept"], cnt["fwd"]), t0
        hi_changed = int(changed.max())
        # inner rounds with frozen P
        for inner in range(20):
            for r in changed[::-1]:
                t = t0 + int(r)
                Psave, psave = W["P"][t].copy(), W["p"][t].copy()
                orc.ctrlbox_backward(prob, act, lo, hi, t, t, W)
                W["P"][t], W["p"][t] = Psave, psave
                cnt["frozen"] += 1
            policy_rollout(u, clipped=clip)
            changed = update()
            if changed.size == 0:
                break
            hi_changed = max(hi_changed, int(changed.max()))
        t_dirty = t0 + hi_changed
    return s, u, mu, (outer, -1, cnt["swept"], cnt["fwd"]), t0


def descent(system, d, strategy, kind):
    At, Bt, ct = d["At"], d["Bt"], d["ct"]
    T, n, m = Bt.shape
    lo, hi = (d["du_lo"], d["du_hi"]) if kind == "rel" else (d["u_lo"], d["u_hi"])
    idx = system.indices_u_into_x
    prob = orc.quasistatic_ctrl_problem(At, Bt, ct, d["Q"], d["Qd"], d["R"], d["xd"], kind)
    W = orc.ctrlbox_workspace(prob)
    a0 = np.sign(d["act"]).astype(int)
    act = np.where((a0 < 0) & np.isfinite(lo), -1, np.where((a0 > 0) & np.isfinite(hi), 1, 0))
    x_new, u_new = np.zeros((T + 1, n)), np.zeros((T, m))
    x_new[0] = d["x0"]
    stats, valid_from = [], T
    u = np.zeros((T, m))
    for t in range(T):
        s_t = np.concatenate([x_new[t], x_new[t][idx]])
        if t == 0 and not act.any():
            orc.ctrlbox_saturated_start(prob, s_t, lo, hi, act, W, 1e-10)
        s, u, mu, st, valid_from = solve(prob, s_t, t, lo, hi, u, act, W, valid_from, strategy)
        stats.append(st)
        ctl = np.clip(u[t], lo[t], hi[t])
        u_new[t] = ctl if kind == "abs" else s_t[n:] + ctl
        x_new[t + 1] = system.dynamics(x_new[t], u_new[t])
    return x_new, u_new, stats


def main():
    path = sys.argv[1]
    strategies = sys.argv[2:] or ["base"]
    d = dict(np.load(path, allow_pickle=True))
    kind = str(d["kind"])
    system = orc.BoxPivotOracle(0.1) if "box_pivoting" in path else orc.PlanarHandOracle(0.1)
    ref = None
    for sname in strategies:
        x_new, u_new, stats = descent(system, d, sname, kind)
        st = np.array(stats)
        failed = [(t, int(a), int(b)) for t, (a, b, _, _) in enumerate(stats) if b != 0]
        print("%-10s iterations %4d (primal-dual %d + primal %d)  backward steps %6d  forward steps %6d  tails into phase 2: %s" % (
            sname, int(st[:, 0].sum() + np.abs(st[:, 1]).sum()), int(st[:, 0].sum()), int(np.abs(st[:, 1]).sum()),
            int(st[:, 2].sum()), int(st[:, 3].sum()), failed), flush=True)
        fz = sum(c.get("frozen", 0) for c in FROZEN_TOTAL)
        if fz:
            print("           single-step updates with the cost-to-go frozen: %d" % fz)
        del FROZEN_TOTAL[:]
        if ref is None:
            ref = u_new
        else:
            print("           |u_new - first strategy's| = %.2e" % np.abs(u_new - ref).max())


if __name__ == "__main__":
    main()
