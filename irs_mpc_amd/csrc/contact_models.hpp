// Planar quasi-dynamic contact dynamics as device functors.
//
// The reference's contact examples (planar_hand, box_pivoting, ...) step an EXTERNAL
// simulator, pangtao22/quasistatic_simulator (not vendored; call sites
// irs_lqr/quasistatic_dynamics.py:136-164).  Its time-stepping scheme is Anitescu's convex
// relaxation of quasi-dynamic contact (Pang & Tedrake 2021): each step solves
//
//     min_dq  1/2 dq_u' (M_u/h^2) dq_u - tau_u' dq_u  +  1/2 (q_a + dq_a - u)' K_a (q_a + dq_a - u)
//     s.t.    phi_i + (n_i + mu d_ij)' J_i dq >= 0      for every contact i, friction generator j
//
// (unactuated DOFs regularised by their mass -- `is_quasi_dynamic=True`; actuated DOFs are
// stiffness-controlled towards the command u; `nd_per_contact=2` generators in the plane), and
// q+ = q + dq.  The only in-tree statement of the model is the 1-D toy of
// examples/box_pushing/analysis/box_on_box.py:11-20 (w1 = m/(m+h^2 k), w2 = h^2 k/(m+h^2 k)),
// which this QP reproduces.  Geometry constants come from the analysis plotters
// (examples/planar_hand/analysis/planar_hand_analysis.py:33-101) and the set-up files.
// PARITY UNPINNED: the simulator's model files (SDF/YAML: masses, friction) are not in the
// tree, so these functors are checked against their own NumPy restatement only.
//
// The QP has a diagonal Hessian, so its dual is a small non-negative QP in the contact
// multipliers, lam >= 0:  min 1/2 lam' W lam + r' lam,  W = J D^-1 J',  r = phi - J D^-1 b,
// solved by projected Gauss-Seidel; dq = D^-1 (J' lam - b).
#pragma once
#include <type_traits>

#include "dual.hpp"

template <typename S>
IRS_HD S irs_max0(const S& a) { return irs_value(a) > 0 ? a : S(a * (typename scalar_of<S>::type)(0)); }


// Shared tail of every quasi-dynamic contact step:  min_dq 1/2 dq'D dq + b'dq  s.t.  phi + J dq >= 0
// (D diagonal), solved through its dual  min_{lam >= 0} 1/2 lam'W lam + r'lam,  W = J D^-1 J',
// r = phi - J D^-1 b, by `iters` projected Gauss-Seidel sweeps; qn = q + D^-1 (J'lam - b).
// PGS runs in residual form: g = r + W lam is kept up to date, so one update is
//   lam_i <- max(lam_i - omega g_i / W_ii, 0),  g += W[:,i] (lam_i_new - lam_i_old)
// -- a 4-deep dependent chain and NC independent FMAs (NC/2 packed ones in f32) instead of an
// NC-term dot product per update.  Fixed sweep count: deterministic, branch-free per sample.
// Over-relaxation of the projected sweeps (projected SOR; 1 = Gauss-Seidel).  W is near-singular when
// several contacts load the same body (8 rows over 7 dofs on the planar hand), and plain sweeps then
// crawl: measured on the planar hand's settled grasp (u-noise std 0.05), the error of 50 sweeps against
// the exactly solved QP drops from 1e-2 (omega = 1) to 7e-4 (1.5), of 100 sweeps from 2e-3 to 1e-6; at
// std 0.3 the 90th percentile drops 10-40x.  Free: the factor is folded into 1 / W_ii.
constexpr double kContactPgsOmega = 1.5;
constexpr int kContactExactWarmSweeps = 32;    // projected sweeps that guess the active set of the exact solve
#ifndef IRS_TRY_SWEEPS
#define IRS_TRY_SWEEPS 32
#endif
constexpr int kContactTrySweeps = IRS_TRY_SWEEPS;   // ... of the sample pass's first attempt (irs_contact_qp_dual_exact_try)

// 1/x: the hardware estimate (1 ulp) for f32 lanes, a true divide otherwise -- the dual active-set loop divides
// 2 NC times per step, and a correctly rounded f32 divide is ~10 instructions
IRS_HD float irs_rcp_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
// f64 lanes (the nominal step, rollouts, the descent's true-dynamics step: one lane, a latency chain): the
// hardware estimate + two Newton steps (~1 ulp) instead of a correctly rounded divide (~40 dependent
// instructions, 8 + 2 NC times per active-set step)
IRS_HD double irs_rcp_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// Active-set polish after the sweeps.  The sweeps identify the active set I = {lam_i > 0} long before they
// converge on it (W is near-singular when several contacts load one body); one exact solve ON that set,
// lam_I += -W_II^-1 g_I (masked LDL' in row order, as in irs_contact_qp_grad), lands on the QP's optimum
// whenever I is right.  It is accepted only if it is the optimum -- multipliers >= 0 on I, slacks >= 0 off
// I, the active slacks solved to zero -- otherwise the swept multipliers stand.  Measured on the planar
// hand (50 sweeps, omega 1.5): 96-99.6 % of the samples accepted and then exact to rounding;
// the share that is more than 1e-5 off the exactly solved QP falls from 14-32 % to 0.4-4 %.
template <typename T, int NC>
IRS_HD void irs_contact_qp_polish(const T (*W)[NC], const T* r, const T* g, T* lam) {
    const T tol_rel = T(1e-6);
    const T piv_rel = sizeof(T) == 4 ? T(1e-5) : T(1e-7);
    T M_[NC][NC], inv[NC], dl[NC];
    bool in[NC];
    T scale = T(1e-30);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        in[i] = lam[i] > T(0);
        scale = fmax(scale, fabs(r[i]));
#pragma unroll
        for (int j = 0; j <= i; ++j) M_[i][j] = W[i][j];
    }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const T dj = M_[j][j];
        const bool piv = dj > piv_rel * W[j][j];
        ok = ok && (piv || !in[j]);                       // a dependent row in I: no polish
        inv[j] = (in[j] && piv) ? irs_rcp_fast(dj) : T(0);
#pragma unroll
        for (int i = j + 1; i < NC; ++i) M_[j][i] = M_[i][j] * inv[j];
#pragma unroll
        for (int i = j + 1; i < NC; ++i)
#pragma unroll
            for (int k = j + 1; k <= i; ++k) M_[i][k] = M_[i][k] - M_[j][i] * M_[k][j];
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        T y = in[j] ? -g[j] : T(0);
#pragma unroll
        for (int k = 0; k < j; ++k) y = y - M_[k][j] * dl[k];
        dl[j] = y;
    }
#pragma unroll
    for (int j = NC - 1; j >= 0; --j) {
        T y = dl[j] * inv[j];
#pragma unroll
        for (int i = j + 1; i < NC; ++i) y = y - M_[j][i] * dl[i];
        dl[j] = y;
    }
    const T tolv = tol_rel * scale;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        T gn = g[i];
#pragma unroll
        for (int j = 0; j < NC; ++j) gn = gn + W[i][j] * dl[j];
        ok = ok && (in[i] ? (lam[i] + dl[i] >= T(0) && fabs(gn) <= tolv) : gn >= -tolv);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) lam[i] = ok ? (in[i] ? lam[i] + dl[i] : T(0)) : lam[i];
}

template <typename S, int NX, int NC>
IRS_HD void irs_contact_qp_dual(const typename scalar_of<S>::type* Dinv, const S* b, const S (*J)[NX],
                                const S* phi, int iters, S (*W)[NC], S* lam) {
    using T = typename scalar_of<S>::type;
    static_assert(NC % 2 == 0, "friction generators come in pairs");
    S r[NC], invW[NC], JD[NC][NX], Db[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) Db[k] = b[k] * Dinv[k];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
#pragma unroll
        for (int k = 0; k < NX; ++k) JD[i][k] = J[i][k] * Dinv[k];
        S ri = phi[i];
#pragma unroll
        for (int k = 0; k < NX; ++k) ri = ri - J[i][k] * Db[k];
        r[i] = ri;
        lam[i] = ri * T(0);
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            S w = JD[i][0] * J[j][0];
#pragma unroll
            for (int k = 1; k < NX; ++k) w = w + JD[i][k] * J[j][k];
            W[i][j] = w;
            W[j][i] = w;
        }
        invW[i] = S(T(kContactPgsOmega)) / W[i][i];      // the relaxation factor rides in the reciprocal
    }
    if constexpr (std::is_same<S, float>::value) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 Wc[NC][NC / 2], g2[NC / 2];
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int k = 0; k < NC / 2; ++k) Wc[i][k] = f2{W[2 * k][i], W[2 * k + 1][i]};
#pragma unroll
        for (int k = 0; k < NC / 2; ++k) g2[k] = f2{r[2 * k], r[2 * k + 1]};
        // two sweeps per trip: the multipliers ping-pong between two register sets, which spares the copy of
        // every old lam_i that a single-sweep loop body needs at its back edge (8 of its 67 instructions)
#pragma unroll 2
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const float gi = (i & 1) ? g2[i / 2].y : g2[i / 2].x;
                const float nw = fmaxf(fmaf(-gi, invW[i], lam[i]), 0.f);
                const float dl = nw - lam[i];
                lam[i] = nw;
                const f2 d2 = f2{dl, dl};
#pragma unroll
                for (int k = 0; k < NC / 2; ++k) g2[k] = Wc[i][k] * d2 + g2[k];
            }
        }
        // the polish reads W back out of the packed copy the sweeps used, so that W itself need not stay in
        // registers across them (a caller that has no further use for W then never holds both)
        float g[NC], Wl[NC][NC];
#pragma unroll
        for (int k = 0; k < NC / 2; ++k) { g[2 * k] = g2[k].x; g[2 * k + 1] = g2[k].y; }
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int i = 0; i < NC; ++i) Wl[a][i] = (a & 1) ? Wc[i][a / 2].y : Wc[i][a / 2].x;
        irs_contact_qp_polish<float, NC>(Wl, r, g, lam);
    } else {
        S g[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) g[i] = r[i];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const S nw = irs_max0(S(lam[i] - g[i] * invW[i]));
                const S dl = nw - lam[i];
                lam[i] = nw;
#pragma unroll
                for (int j = 0; j < NC; ++j) g[j] = g[j] + W[j][i] * dl;
            }
        }
        if constexpr (std::is_arithmetic<S>::value) irs_contact_qp_polish<S, NC>(W, r, g, lam);
    }
}

// ---- EXACT dual solve: the Goldfarb-Idnani dual active-set method written in the dual variables --------
// The primal QP has the diagonal Hessian D, so the Schur complement of an active set A is W_AA.  Start from
// lam = 0 (the unconstrained primal optimum) and repeat: p = the most violated row (slack g_p = (r + W lam)_p
// < 0) outside A; step along  d lam_A = -rho, d lam_p = +1,  rho = W_AA^-1 W_Ap  -- the active slacks stay at
// zero, g_p rises at rate z = W_pp - W_pA rho -- until g_p = 0 (full step: p joins A) or some lam_i in A
// reaches zero first (partial step: i leaves A, p stays the candidate).  z = 0 marks a row that depends on
// A: only partial steps are possible.  Every step increases the dual objective: finite, no cycling, no
// regularisation; dependent rows never enter A, so W_AA stays positive definite.  (Projected sweeps crawl on
// exactly these problems: W is near-singular when several contacts load one body -- DESIGN.md 7.)
// SIMT form: no lane ever indexes a register array by a run-time value -- the candidate row and the
// blocking row are one-hot vectors, the factorisation of W_AA is the masked LDL' of irs_contact_qp_grad --
// and the loop is wave-uniform: it ends when every lane of the wave is done (cap 4 NC steps; a lane that
// hits the cap keeps its last multipliers, which are dual feasible).  Restated in oracle/irs_oracle.py
// (_ContactQPOracle._dual_exact, which starts every sample from lam = 0: the primal solution is unique, so
// the warm start below changes the path, not the answer).  T = float or double.
template <typename T>
IRS_HD bool irs_wave_all(bool v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __all(v);
#else
    return v;
#endif
}

// `warm` (optional, in/out): bit i = row i of the active set.  In: a guess that REPLACES the warm-up sweeps
// (~0u = none: sweep) -- consecutive steps of one trajectory bind nearly the same contacts, and a one-lane
// f64 step is a latency chain in which the 32 sweeps are 2/3 of the time; out: the set the solve ended on.
// The guess only shortens the path: the primal solution is unique.
template <typename T, int NX, int NC>
IRS_HD void irs_contact_qp_dual_exact(const T* Dinv, const T* b, const T (*J)[NX], const T* phi, T (*W)[NC],
                                      T* lam, unsigned* warm = nullptr) {
    constexpr T kBig = T(3.0e38);
    const T tol_rel = sizeof(T) == 4 ? T(1e-6) : T(1e-10);
    const T piv_rel = sizeof(T) == 4 ? T(1e-5) : T(1e-7);
    T g[NC], Wd[NC];
    bool act[NC];
    {
        T JD[NC][NX], Db[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) Db[k] = b[k] * Dinv[k];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
#pragma unroll
            for (int k = 0; k < NX; ++k) JD[i][k] = J[i][k] * Dinv[k];
            T ri = phi[i];
#pragma unroll
            for (int k = 0; k < NX; ++k) ri = ri - J[i][k] * Db[k];
            g[i] = ri;
            lam[i] = T(0);
            act[i] = false;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                T w = JD[i][0] * J[j][0];
#pragma unroll
                for (int k = 1; k < NX; ++k) w = w + JD[i][k] * J[j][k];
                W[i][j] = w;
                W[j][i] = w;
            }
            Wd[i] = W[i][i];
        }
    }
    T scale = T(1e-30);
#pragma unroll
    for (int i = 0; i < NC; ++i) scale = fmax(scale, fabs(g[i]));
    const T tolv = tol_rel * scale;
    // ---- warm start.  The method may start from ANY pair (A, lam) with lam_A = -W_AA^-1 r_A >= 0 (the
    // optimum of the QP restricted to A, multipliers of the right sign); a few over-relaxed projected sweeps
    // guess A, and rows whose restricted multiplier comes out negative are released until the pair is
    // valid.  From there the loop below typically needs 0-3 steps instead of 8-13 from lam = 0, and a wave
    // runs as long as its slowest lane.  A lane whose guess cannot be repaired starts from lam = 0.
    {
        T r[NC], invw[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) { r[i] = g[i]; invw[i] = T(kContactPgsOmega) * irs_rcp_fast(Wd[i]); }
        const bool guessed = warm != nullptr && *warm != ~0u;
        if (guessed) {
#pragma unroll
            for (int i = 0; i < NC; ++i) act[i] = ((*warm >> i) & 1u) != 0u;
        } else {
#pragma unroll 2
            for (int sw = 0; sw < kContactExactWarmSweeps; ++sw) {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const T nw = fmax(lam[i] - g[i] * invw[i], T(0));
                    const T dl = nw - lam[i];
                    lam[i] = nw;
#pragma unroll
                    for (int j = 0; j < NC; ++j) g[j] = g[j] + W[j][i] * dl;
                }
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) act[i] = lam[i] > T(0);
        }
        bool valid = false;
        for (int round = 0; round < 3; ++round) {
            if (irs_wave_all<T>(valid)) break;
            T M_[NC][NC], inv[NC], y[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) M_[i][j] = W[i][j];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const T dj = M_[j][j];
                const bool ok = act[j] && dj > piv_rel * Wd[j];
                if (!valid) act[j] = ok;                              // a dependent row leaves the guess
                inv[j] = ok ? irs_rcp_fast(dj) : T(0);
#pragma unroll
                for (int i = j + 1; i < NC; ++i) M_[j][i] = M_[i][j] * inv[j];
#pragma unroll
                for (int i = j + 1; i < NC; ++i)
#pragma unroll
                    for (int k = j + 1; k <= i; ++k) M_[i][k] = M_[i][k] - M_[j][i] * M_[k][j];
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                T v = act[j] ? -r[j] : T(0);
#pragma unroll
                for (int k = 0; k < j; ++k) v = v - M_[k][j] * y[k];
                y[j] = v;
            }
#pragma unroll
            for (int j = NC - 1; j >= 0; --j) {
                T v = y[j] * inv[j];
#pragma unroll
                for (int i = j + 1; i < NC; ++i) v = v - M_[j][i] * y[i];
                y[j] = v;
            }
            if (!valid) {
                bool all_pos = true;
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const bool neg = act[i] && !(y[i] > T(0));
                    all_pos = all_pos && !neg;
                    lam[i] = act[i] ? y[i] : T(0);
                    if (neg) act[i] = false;
                }
                valid = all_pos;
            }
        }
        // slacks of the pair; an unrepaired guess falls back to the cold start
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (!valid) { lam[i] = T(0); act[i] = false; }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            T s = r[i];
#pragma unroll
            for (int j = 0; j < NC; ++j) s = s + W[i][j] * lam[j];
            g[i] = act[i] ? T(0) : s;
        }
    }
    int p = -1;                    // candidate row, -1 = none
    bool done = false;
    for (int it = 0; it < 4 * NC; ++it) {
        if (!done && p < 0) {
            T vmin = kBig;
            int c = 0;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const T v = act[i] ? kBig : g[i];
                if (v < vmin) { vmin = v; c = i; }
            }
            if (vmin >= -tolv) done = true;
            else p = c;
        }
        if (irs_wave_all<T>(done)) break;
        // candidate row as a one-hot vector; its column of W, its diagonal entry and its slack
        T e[NC], wp[NC];
        T wpp = T(0), gp = T(0);
#pragma unroll
        for (int i = 0; i < NC; ++i) e[i] = (!done && i == p) ? T(1) : T(0);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NC; ++j) s = s + W[i][j] * e[j];
            wp[i] = s;
            wpp = wpp + Wd[i] * e[i];
            gp = gp + g[i] * e[i];
        }
        // rho = W_AA^-1 W_Ap: masked LDL' in row order on a copy (unit lower factor in the upper triangle)
        T M_[NC][NC], inv[NC], rho[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) M_[i][j] = W[i][j];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const T dj = M_[j][j];
            const bool ok = act[j] && dj > piv_rel * Wd[j];
            inv[j] = ok ? irs_rcp_fast(dj) : T(0);
#pragma unroll
            for (int i = j + 1; i < NC; ++i) M_[j][i] = M_[i][j] * inv[j];
#pragma unroll
            for (int i = j + 1; i < NC; ++i)
#pragma unroll
                for (int k = j + 1; k <= i; ++k) M_[i][k] = M_[i][k] - M_[j][i] * M_[k][j];
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            T y = act[j] ? wp[j] : T(0);
#pragma unroll
            for (int k = 0; k < j; ++k) y = y - M_[k][j] * rho[k];
            rho[j] = y;
        }
#pragma unroll
        for (int j = NC - 1; j >= 0; --j) {
            T y = rho[j] * inv[j];
#pragma unroll
            for (int i = j + 1; i < NC; ++i) y = y - M_[j][i] * rho[i];
            rho[j] = y;
        }
        // step lengths
        T zp = wpp;
#pragma unroll
        for (int i = 0; i < NC; ++i) zp = zp - (act[i] ? wp[i] * rho[i] : T(0));
        const bool full_ok = zp > piv_rel * wpp;
        const T t2 = full_ok ? -gp * irs_rcp_fast(zp) : kBig;
        T t1 = kBig;
        int kb = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const bool cand = act[i] && rho[i] > T(0);
            const T q = cand ? lam[i] * irs_rcp_fast(rho[i]) : kBig;
            if (q < t1) { t1 = q; kb = i; }
        }
        const T tmin = fmin(t1, t2);
        if (!done && !(tmin < kBig)) done = true;          // no step possible: infeasible primal, keep lam
        const T t = done ? T(0) : tmin;
        // lam_A -= t rho, lam_p += t, g += t (W_p - W rho)
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            T s = wp[i];
#pragma unroll
            for (int j = 0; j < NC; ++j) s = s - W[i][j] * rho[j];
            g[i] = g[i] + t * s;
            lam[i] = fmax(lam[i] - t * rho[i] + t * e[i], T(0));
        }
        if (!done) {
            const bool full = t2 <= t1;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (full) {
                    if (i == p) act[i] = true;
                } else if (i == kb) {
                    act[i] = false;
                    lam[i] = T(0);
                }
            }
            if (full) p = -1;
        }
    }
    if (warm != nullptr) {
        unsigned mk = 0u;
#pragma unroll
        for (int i = 0; i < NC; ++i) mk |= act[i] ? (1u << i) : 0u;
        *warm = mk;
    }
}

// FIRST ATTEMPT of the exact solve, for the sample pass (smooth.hip): the warm-up sweeps, ONE restricted solve
// on the set they guess, and the optimality test -- no loop whose trip count depends on the data.  Returns true
// when the guess IS the optimum (lam = the exact multipliers; ~90 % of the benchmark's samples); otherwise
// `mask` holds the guess with the wrong-signed rows released, a valid `warm` argument for
// irs_contact_qp_dual_exact, which finishes the sample.  Why two functions: inside one wave the data-dependent
// part (repair rounds, active-set steps) runs as long as its slowest lane needs -- 1 lane in 20 takes a step,
// every wave paid for two.  The sample pass therefore parks unfinished samples in a per-wave LDS queue and
// finishes them 64 at a time, all lanes busy.  Same arithmetic as the first round of the full method, so a
// sample that passes here gets the multipliers the full method would give it.
template <typename T, int NX, int NC>
IRS_HD bool irs_contact_qp_dual_exact_try(const T* Dinv, const T* b, const T (*J)[NX], const T* phi, T (*W)[NC],
                                          T* lam, unsigned* mask) {
    const T tol_rel = sizeof(T) == 4 ? T(1e-6) : T(1e-10);
    const T piv_rel = sizeof(T) == 4 ? T(1e-5) : T(1e-7);
    T g[NC], r[NC], Wd[NC], invw[NC];
    bool act[NC];
    {
        T JD[NC][NX], Db[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) Db[k] = b[k] * Dinv[k];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
#pragma unroll
            for (int k = 0; k < NX; ++k) JD[i][k] = J[i][k] * Dinv[k];
            T ri = phi[i];
#pragma unroll
            for (int k = 0; k < NX; ++k) ri = ri - J[i][k] * Db[k];
            g[i] = ri;
            r[i] = ri;
            lam[i] = T(0);
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                T w = JD[i][0] * J[j][0];
#pragma unroll
                for (int k = 1; k < NX; ++k) w = w + JD[i][k] * J[j][k];
                W[i][j] = w;
                W[j][i] = w;
            }
            Wd[i] = W[i][i];
            invw[i] = T(kContactPgsOmega) * irs_rcp_fast(Wd[i]);
        }
    }
    T scale = T(1e-30);
#pragma unroll
    for (int i = 0; i < NC; ++i) scale = fmax(scale, fabs(r[i]));
    const T tolv = tol_rel * scale;
#pragma unroll 2
    for (int sw = 0; sw < kContactTrySweeps; ++sw) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const T nw = fmax(lam[i] - g[i] * invw[i], T(0));
            const T dl = nw - lam[i];
            lam[i] = nw;
#pragma unroll
            for (int j = 0; j < NC; ++j) g[j] = g[j] + W[j][i] * dl;
        }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) act[i] = lam[i] > T(0);
    // the restricted optimum on the guess: masked LDL' in row order (a dependent row leaves the guess)
    T M_[NC][NC], inv[NC], y[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) M_[i][j] = W[i][j];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const T dj = M_[j][j];
        const bool ok = act[j] && dj > piv_rel * Wd[j];
        act[j] = ok;
        inv[j] = ok ? irs_rcp_fast(dj) : T(0);
#pragma unroll
        for (int i = j + 1; i < NC; ++i) M_[j][i] = M_[i][j] * inv[j];
#pragma unroll
        for (int i = j + 1; i < NC; ++i)
#pragma unroll
            for (int k = j + 1; k <= i; ++k) M_[i][k] = M_[i][k] - M_[j][i] * M_[k][j];
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        T v = act[j] ? -r[j] : T(0);
#pragma unroll
        for (int k = 0; k < j; ++k) v = v - M_[k][j] * y[k];
        y[j] = v;
    }
#pragma unroll
    for (int j = NC - 1; j >= 0; --j) {
        T v = y[j] * inv[j];
#pragma unroll
        for (int i = j + 1; i < NC; ++i) v = v - M_[j][i] * y[i];
        y[j] = v;
    }
    // the tests as min-reductions (one compare each at the end): a chain of `good && ...` over 2 NC vector compares
    // serialises on the scalar unit -- measured 1 590 cycles for ~110 instructions
    unsigned mk = 0u;
    T ymin = T(3.0e38);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const bool neg = act[i] && !(y[i] > T(0));
        ymin = fmin(ymin, act[i] ? y[i] : T(3.0e38));
        lam[i] = act[i] ? y[i] : T(0);
        mk |= (act[i] && !neg) ? (1u << i) : 0u;
    }
    // slacks off the set: all >= -tol <=> optimal
    T smin = T(3.0e38);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        T s = r[i];
#pragma unroll
        for (int j = 0; j < NC; ++j) s = s + W[i][j] * lam[j];
        smin = fmin(smin, act[i] ? T(3.0e38) : s);
    }
    const bool good = ymin > T(0) && smin >= -tolv;
    *mask = mk;
    return good;
}

template <typename S, int NX, int NC>
IRS_HD void irs_contact_qp_primal(const S* q, const typename scalar_of<S>::type* Dinv, const S* b,
                                  const S (*J)[NX], const S* lam, S* qn) {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        S f = -b[k];
#pragma unroll
        for (int i = 0; i < NC; ++i) f = f + J[i][k] * lam[i];
        qn[k] = q[k] + f * Dinv[k];
    }
}

template <typename S, int NX, int NC>
IRS_HD void irs_contact_qp_step(const S* q, const typename scalar_of<S>::type* Dinv, const S* b,
                                const S (*J)[NX], const S* phi, int iters, S* qn) {
    S W[NC][NC], lam[NC];
    irs_contact_qp_dual<S, NX, NC>(Dinv, b, J, phi, iters, W, lam);
    irs_contact_qp_primal<S, NX, NC>(q, Dinv, b, J, lam, qn);
}

// Derivative of the step through its ACTIVE constraints, contact geometry (J) held fixed -- what the
// reference reads from the simulator after `step(..., requires_grad=True, grad_from_active_constraints=True)`:
// q_sim.get_Dq_nextDq() | q_sim.get_Dq_nextDqa_cmd() (irs_lqr/quasistatic_dynamics.py:143-164, 184-191).
// With I = {i : lam_i W_ii > kContactActiveTol}, W_II = J_I D^-1 J_I':
//     S  = J_I' W_II^+ J_I    (b enters linearly:                    d dq / d b     = -D^-1 + D^-1 S D^-1)
//     Sn = J_I' W_II^+ Jn_I   (phi_i is a gap, d phi_i / d q = the NORMAL row Jn_i = mean of the contact's
//                              two generator rows:                   d dq / d phi_I = -D^-1 J_I' W_II^+)
// and b_a = K (q_a - u), D_aa = K on the actuated dofs `act[j]`, so in the internal coordinate order
//     B = E_a - D^-1 S[:, a],        A[:, l] = e_l - [l = a_j] B[:, j] - D^-1 Sn[:, l].
// W_II is factorised IN PLACE (W is destroyed) by a masked LDL' in row order: inactive rows get a zero
// inverse pivot, and so does an active row that depends on earlier ones (pivot < kContactPivotTol W_ii);
// the projector S does not depend on which dependent row is dropped.  Pinned by the simulator's own
// Jacobians, examples/box_pushing/analysis/dxdu_quasistatic.npy (tests/).  T = float or double only.
constexpr double kContactActiveTol = 1e-7;
constexpr double kContactPivotTol = 1e-5;

template <typename T, int NX, int NC, int NA, bool WITH_A>
IRS_HD void irs_contact_qp_grad(const T* Dinv, const T (*J)[NX], T (*W)[NC], const T* lam, const int* act,
                                T (*Bint)[NA], T (*Aint)[NX]) {
    constexpr int NR = NA + (WITH_A ? NX : 0);       // right-hand sides: J[:, act] and (for A) Jn
    T inv[NC], Y[NC][NR];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
#pragma unroll
        for (int c = 0; c < NA; ++c) Y[i][c] = J[i][act[c]];
        if constexpr (WITH_A) {
#pragma unroll
            for (int l = 0; l < NX; ++l) Y[i][NA + l] = T(0.5) * (J[i & ~1][l] + J[i | 1][l]);
        }
    }
    // masked LDL': unit lower factor kept in the UPPER triangle (W[j][i] = L_ij), the lower one holds the
    // running Schur complement
    T Wd[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) Wd[j] = W[j][j];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const T dj = W[j][j];
        const bool ok = (lam[j] * Wd[j] > T(kContactActiveTol)) && (dj > T(kContactPivotTol) * Wd[j]);
        inv[j] = ok ? T(1) / dj : T(0);
#pragma unroll
        for (int i = j + 1; i < NC; ++i) W[j][i] = W[i][j] * inv[j];
#pragma unroll
        for (int i = j + 1; i < NC; ++i)
#pragma unroll
            for (int k = j + 1; k <= i; ++k) W[i][k] = W[i][k] - W[j][i] * W[k][j];
    }
#pragma unroll
    for (int c = 0; c < NR; ++c) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            T y = Y[j][c];
#pragma unroll
            for (int k = 0; k < j; ++k) y = y - W[k][j] * Y[k][c];
            Y[j][c] = y;
        }
#pragma unroll
        for (int j = NC - 1; j >= 0; --j) {
            T y = Y[j][c] * inv[j];
#pragma unroll
            for (int i = j + 1; i < NC; ++i) y = y - W[j][i] * Y[i][c];
            Y[j][c] = y;
        }
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) {
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            T s = T(0);
#pragma unroll
            for (int i = 0; i < NC; ++i) s = s + J[i][k] * Y[i][c];
            Bint[k][c] = (k == act[c] ? T(1) : T(0)) - Dinv[k] * s;
        }
    }
    if constexpr (WITH_A) {
#pragma unroll
        for (int k = 0; k < NX; ++k)
#pragma unroll
            for (int l = 0; l < NX; ++l) {
                T s = T(0);
#pragma unroll
                for (int i = 0; i < NC; ++i) s = s + J[i][k] * Y[i][NA + l];
                T a = (k == l ? T(1) : T(0)) - Dinv[k] * s;
#pragma unroll
                for (int c = 0; c < NA; ++c)
                    if (act[c] == l) a = a - Bint[k][c];
                Aint[k][l] = a;
            }
    }
}

// contact models with `static constexpr bool EXACT = true` solve the step QP exactly (irs_contact_qp_dual_exact)
template <class M, class = void>
struct irs_contact_exact : std::false_type {};
template <class M>
struct irs_contact_exact<M, std::void_t<decltype(M::EXACT)>> : std::integral_constant<bool, M::EXACT> {};

// One step of contact model M (its `assemble` builds the QP in the internal coordinate order).
template <class M, typename S>
IRS_HD void irs_contact_step(const ModelParams& p, const S* x_ext, const S* u, S* xn_ext, unsigned* warm = nullptr) {
    using T = typename scalar_of<S>::type;
    constexpr int NX = M::NX, NC = M::NC;
    S q[NX], qn[NX], b[NX], J[NC][NX], phi[NC];
    T Dinv[NX];
    const int iters = M::template assemble<S>(p, x_ext, u, q, Dinv, b, J, phi);
    if constexpr (irs_contact_exact<M>::value) {
        S W[NC][NC], lam[NC];
        irs_contact_qp_dual_exact<S, NX, NC>(Dinv, b, J, phi, W, lam, warm);
        irs_contact_qp_primal<S, NX, NC>(q, Dinv, b, J, lam, qn);
    } else {
        irs_contact_qp_step<S, NX, NC>(q, Dinv, b, J, phi, iters, qn);
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) xn_ext[M::perm(k)] = qn[k];
}

// The step and its active-set derivative in the reference's x order: Bext (NX x NU, row-major) and,
// WITH_A, Aext (NX x NX).
// `active_mask` (optional): bit i set = contact row i is in the active set the derivative is taken through
// (lam_i W_ii > kContactActiveTol) -- diagnostics / tests only.
template <class M, typename T, bool WITH_A>
IRS_HD void irs_contact_step_grad(const ModelParams& p, const T* x_ext, const T* u, T* xn_ext, T* Bext, T* Aext,
                                  unsigned* active_mask = nullptr, unsigned* warm = nullptr) {
    constexpr int NX = M::NX, NC = M::NC, NU = M::NU;
    T q[NX], qn[NX], b[NX], J[NC][NX], phi[NC], Dinv[NX], W[NC][NC], lam[NC];
    const int iters = M::template assemble<T>(p, x_ext, u, q, Dinv, b, J, phi);
    if constexpr (irs_contact_exact<M>::value) irs_contact_qp_dual_exact<T, NX, NC>(Dinv, b, J, phi, W, lam, warm);
    else irs_contact_qp_dual<T, NX, NC>(Dinv, b, J, phi, iters, W, lam);
    irs_contact_qp_primal<T, NX, NC>(q, Dinv, b, J, lam, qn);
    if (active_mask != nullptr) {
        unsigned mk = 0u;
#pragma unroll
        for (int i = 0; i < NC; ++i) mk |= (lam[i] * W[i][i] > T(kContactActiveTol)) ? (1u << i) : 0u;
        *active_mask = mk;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) xn_ext[M::perm(k)] = qn[k];
    int act[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) act[j] = M::act(j);
    T Bint[NX][NU], Aint[WITH_A ? NX : 1][NX];
    irs_contact_qp_grad<T, NX, NC, NU, WITH_A>(Dinv, J, W, lam, act, Bint, Aint);
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
        for (int j = 0; j < NU; ++j) Bext[M::perm(k) * NU + j] = Bint[k][j];
    if constexpr (WITH_A) {
#pragma unroll
        for (int k = 0; k < NX; ++k)
#pragma unroll
            for (int l = 0; l < NX; ++l) Aext[M::perm(k) * NX + M::perm(l)] = Aint[k][l];
    }
}

// examples/planar_hand: a disc (radius R) cradled by two 2-link arms with capsule links.
//   x = [xo, ql1, qr1, yo, ql2, qr2, th]   -- the reference's state order (Drake's velocity indices
//       of the plant; examples/planar_hand/analysis/planar_hand_analysis.py:61-67)
//   u = commanded joint angles [ql1, ql2, qr1, qr2]      (indices_u_into_x = 1, 4, 2, 5)
//   internally q = [xo, yo, th, ql1, ql2, qr1, qr2] (object first); PERM maps q's index to x's
//   params = {h, g, mass, R, mu, kp1, kp2, l1, l2, r_link, base_x, pgs_iters}
// Geometry (planar_hand_analysis.py:69-101): bases (-+base_x, 0); the left arm's first joint
// angle is offset by pi; link lengths l1, l2; capsule radius r_link.
struct PlanarHandModel {
    static constexpr int NX = 7, NU = 4, NPARAMS = 12;
    static constexpr int NC = 8;                 // 4 link-disc pairs x 2 friction generators
    // not differentiable by dual numbers (HAS_JACOBIAN = false): its Jacobian is the active-set
    // derivative of the step QP (irs_contact_qp_grad); the sample-pass modes return the decoupled (A,B)
    // of IrsLqrQuasistatic.decouple_AB_matrices (irs_lqr_quasistatic.py:275-284) -- every contact
    // example sets decouple_AB = True, which discards the sampled A anyway
    static constexpr bool HAS_JACOBIAN = false;
    IRS_HD static constexpr int perm(int k) {       // internal index k -> index in the reference's x
        return k == 0 ? 0 : k == 1 ? 3 : k == 2 ? 6 : k == 3 ? 1 : k == 4 ? 4 : k == 5 ? 2 : 5;
    }
    IRS_HD static constexpr int act(int j) { return 3 + j; }      // internal index of the j-th actuated dof
    IRS_HD static int u_into_x(int j) { return perm(act(j)); }
    // the impedance gain of the j-th commanded joint: assemble's b[act(j)] = stiffness(j) * (q[act(j)] - u[j]), the ONLY
    // place u enters the step QP -- lets a caller with a fixed state assemble the geometry once (smooth.hip)
    template <typename T>
    IRS_HD static T stiffness(const ModelParams& p, int j) { return (j & 1) ? T(p.v[6]) : T(p.v[5]); }

    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x_ext, const S* u, S* xn_ext) {
        irs_contact_step<PlanarHandModel, S>(p, x_ext, u, xn_ext);
    }

    // the step QP in the internal order: q, D^-1, b, J, phi; returns the PGS sweep count
    template <typename S>
    IRS_HD static int assemble(const ModelParams& p, const S* x_ext, const S* u, S* q,
                               typename scalar_of<S>::type* Dinv, S* b, S (*J)[NX], S* phi) {
        using T = typename scalar_of<S>::type;
#pragma unroll
        for (int k = 0; k < NX; ++k) q[k] = x_ext[perm(k)];
        const T h = T(p.v[0]), g = T(p.v[1]), mass = T(p.v[2]), R = T(p.v[3]), mu = T(p.v[4]);
        const T kp1 = T(p.v[5]), kp2 = T(p.v[6]), l1 = T(p.v[7]), l2 = T(p.v[8]), rl = T(p.v[9]), bx = T(p.v[10]);
        const int iters = (int)p.v[11];
        // diagonal QP Hessian D and linear term b
        const T inertia = T(0.5) * mass * R * R;
        Dinv[0] = h * h / mass; Dinv[1] = h * h / mass; Dinv[2] = h * h / inertia;
        Dinv[3] = T(1) / kp1; Dinv[4] = T(1) / kp2; Dinv[5] = T(1) / kp1; Dinv[6] = T(1) / kp2;
        b[0] = q[0] * T(0);
        b[1] = q[1] * T(0) + mass * g;           // -tau_u, tau_u = (0, -m g, 0)
        b[2] = q[2] * T(0);
        b[3] = kp1 * (q[3] - u[0]); b[4] = kp2 * (q[4] - u[1]);
        b[5] = kp1 * (q[5] - u[2]); b[6] = kp2 * (q[6] - u[3]);

        // contact rows: J (NC x NX), phi (NC)
#pragma unroll
        for (int arm = 0; arm < 2; ++arm) {
            const T base = arm == 0 ? -bx : bx;
            const S a1 = arm == 0 ? S(q[3] + T(3.14159265358979323846)) : q[5];
            const S a2 = a1 + (arm == 0 ? q[4] : q[6]);
            S s1, c1, s2, c2;
            irs_sincos(a1, s1, c1);
            irs_sincos(a2, s2, c2);
            const S p1x = c1 * l1 + base, p1y = s1 * l1;      // second joint
#pragma unroll
            for (int link = 0; link < 2; ++link) {
                // segment start a, direction (dx,dy), length L
                const S ax = link == 0 ? S(c1 * T(0) + base) : p1x;
                const S ay = link == 0 ? S(s1 * T(0)) : p1y;
                const S dx = link == 0 ? c1 : c2, dy = link == 0 ? s1 : s2;
                const T L = link == 0 ? l1 : l2;
                S rx = q[0] - ax, ry = q[1] - ay;
                S sp = rx * dx + ry * dy;                      // projection on the segment
                const T spv = irs_value(sp);
                if (spv < T(0)) sp = sp * T(0);
                else if (spv > L) sp = sp * T(0) + L;
                const S wx = ax + dx * sp, wy = ay + dy * sp;  // closest point on the axis
                S nx = q[0] - wx, ny = q[1] - wy;
                const S dist = irs_sqrt(nx * nx + ny * ny);
                nx = nx / dist; ny = ny / dist;                // link -> object
                const S gap = dist - (R + rl);
                // contact point on the link surface, arms about the joints
                const S cx = wx + nx * rl, cy = wy + ny * rl;
                const S r1x = cx - base, r1y = cy;             // about the arm base (z x r = (-ry, rx))
                const S r2x = cx - p1x, r2y = cy - p1y;        // about the second joint
                const S tx = -ny, ty = nx;                     // tangent
#pragma unroll
                for (int gen = 0; gen < 2; ++gen) {
                    const int row = (arm * 2 + link) * 2 + gen;
                    const T sgn = gen == 0 ? mu : -mu;
                    const S ex = nx + tx * sgn, ey = ny + ty * sgn;
                    phi[row] = gap;
                    // object: v = (dxo, dyo) + dth * z x (-R n)  ;  z x (-R n) = (R ny, -R nx)
                    J[row][0] = ex;
                    J[row][1] = ey;
                    J[row][2] = (ex * ny - ey * nx) * R;
                    // link point: -(q1' z x r1 [+ q2' z x r2])
                    const S j1 = -(ey * r1x - ex * r1y);
                    const S j2 = link == 1 ? S(-(ey * r2x - ex * r2y)) : S(ex * T(0));
                    const S zero = ex * T(0);
                    J[row][3] = arm == 0 ? j1 : zero;
                    J[row][4] = arm == 0 ? j2 : zero;
                    J[row][5] = arm == 1 ? j1 : zero;
                    J[row][6] = arm == 1 ? j2 : zero;
                }
            }
        }
        return iters;
    }
};


// The planar hand with its step QP solved EXACTLY (irs_contact_qp_dual_exact) instead of by `pgs_iters`
// sweeps; same parameters (pgs_iters is ignored).  A separate model id so that the kernels of the
// sweep-based functor -- the benchmarked ones -- are not touched by the extra code path.
struct PlanarHandExactModel : PlanarHandModel {
    static constexpr bool EXACT = true;
    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x_ext, const S* u, S* xn_ext) {
        irs_contact_step<PlanarHandExactModel, S>(p, x_ext, u, xn_ext);
    }
};

// Rows of a disc ("hand", radius rh, centre (q[3], q[4])) against a square box (half side a, centre
// (q[0], q[1]), angle q[2]; sn, cs = sin / cos of the angle): two friction-cone generators n +- mu t of
// the pair, written to J[row0], J[row0 + 1] and phi.  Nearest point of the outline by clamping in the
// box frame; a disc centre inside or ON the outline (the reference's box_pivoting start) takes the
// nearest face.  q = [xb, yb, th, xh, yh].
template <typename S, typename T>
IRS_HD void irs_hand_box_rows(const S* q, const S& sn, const S& cs, T a, T rh, T mu, S (*J)[5], S* phi, int row0) {
    const S dx = q[3] - q[0], dy = q[4] - q[1];
    const S px = cs * dx + sn * dy, py = -sn * dx + cs * dy;     // hand centre in the box frame
    const T pxv = irs_value(px), pyv = irs_value(py);
    const bool outside = fabs(pxv) > a || fabs(pyv) > a;
    // outside: nearest point = clamp; inside or on the boundary: the nearest face
    const S cxq = irs_select(pxv > a, S(a), irs_select(pxv < -a, S(-a), px));
    const S cyq = irs_select(pyv > a, S(a), irs_select(pyv < -a, S(-a), py));
    const T ddx = a - fabs(pxv), ddy = a - fabs(pyv);           // penetration depths (inside)
    const bool facex = ddx <= ddy;
    const T sgx = pxv >= T(0) ? T(1) : T(-1), sgy = pyv >= T(0) ? T(1) : T(-1);
    S qlx, qly, nlx, nly, dist;
    if (outside) {
        const S ex = px - cxq, ey = py - cyq;
        dist = irs_sqrt(ex * ex + ey * ey);
        nlx = ex / dist; nly = ey / dist;
        qlx = cxq; qly = cyq;
    } else {
        qlx = facex ? S(sgx * a) : px;
        qly = facex ? py : S(sgy * a);
        nlx = S(facex ? sgx : T(0));
        nly = S(facex ? T(0) : sgy);
        dist = S(-(facex ? ddx : ddy));
    }
    const S gap = dist - rh;
    const S nx = cs * nlx - sn * nly, ny = sn * nlx + cs * nly;  // box -> hand, world frame
    const S rx = cs * qlx - sn * qly, ry = sn * qlx + cs * qly;  // contact point - box centre
#pragma unroll
    for (int gen = 0; gen < 2; ++gen) {
        const int row = row0 + gen;
        const T sg = gen == 0 ? mu : -mu;
        const S ex = nx - ny * sg, ey = ny + nx * sg;            // e = n + sg t, t = (-ny, nx)
        phi[row] = gap;
        J[row][0] = -ex;
        J[row][1] = -ey;
        J[row][2] = -(ey * rx - ex * ry);
        J[row][3] = ex;
        J[row][4] = ey;
    }
}

// examples/box_pivoting: a 1 m square box on the ground, pushed / pivoted by a position-controlled
// disc ("hand", radius 0.1; examples/box_pivoting/analysis/box_pivoting_analysis.py:34-72).
//   x = [x_h, x_b, y_h, y_b, th_b]   -- the reference's state order (box_pivoting_analysis.py:53-64)
//   u = commanded hand position [x_h, y_h]              (indices_u_into_x = 0, 2)
//   internally q = [xb, yb, th, xh, yh]
//   params = {h, g, mass, half, mu, kp, r_hand, pgs_iters}   (box_pivoting_setup.py:9-19: Kp = 5e4,
//            h = 0.1, g = 9.81; the box mass and the friction coefficient are in the absent SDF/YAML)
// Contacts (2 friction generators each): the 4 box corners against the ground y = 0, the hand against
// the box (closest point of the square's boundary; inside / on the boundary: the nearest face), the
// hand against the ground.  PARITY UNPINNED, like the planar hand.
struct BoxPivotModel {
    static constexpr int NX = 5, NU = 2, NPARAMS = 8;
    static constexpr int NC = 12;
    static constexpr bool HAS_JACOBIAN = false;
    IRS_HD static constexpr int perm(int k) { return k == 0 ? 1 : k == 1 ? 3 : k == 2 ? 4 : k == 3 ? 0 : 2; }
    IRS_HD static constexpr int act(int j) { return 3 + j; }
    IRS_HD static int u_into_x(int j) { return perm(act(j)); }
    template <typename T>
    IRS_HD static T stiffness(const ModelParams& p, int) { return T(p.v[5]); }      // b[act(j)] = kp (q - u)

    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x_ext, const S* u, S* xn_ext) {
        irs_contact_step<BoxPivotModel, S>(p, x_ext, u, xn_ext);
    }

    template <typename S>
    IRS_HD static int assemble(const ModelParams& p, const S* x_ext, const S* u, S* q,
                               typename scalar_of<S>::type* Dinv, S* b, S (*J)[NX], S* phi) {
        using T = typename scalar_of<S>::type;
        const T h = T(p.v[0]), g = T(p.v[1]), mass = T(p.v[2]), a = T(p.v[3]), mu = T(p.v[4]);
        const T kp = T(p.v[5]), rh = T(p.v[6]);
        const int iters = (int)p.v[7];
#pragma unroll
        for (int k = 0; k < NX; ++k) q[k] = x_ext[perm(k)];
        const T inertia = mass * (T(2) * a) * (T(2) * a) / T(6);     // square plate, side 2a
        Dinv[0] = h * h / mass; Dinv[1] = h * h / mass; Dinv[2] = h * h / inertia;
        Dinv[3] = T(1) / kp; Dinv[4] = T(1) / kp;
        b[0] = S(T(0)); b[1] = S(mass * g); b[2] = S(T(0));
        b[3] = kp * (q[3] - u[0]); b[4] = kp * (q[4] - u[1]);

        S sn, cs;
        irs_sincos(q[2], sn, cs);
        // rows 0..7: box corners (+-a, +-a) against the ground
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const T lx = (c & 1) ? a : -a, ly = (c & 2) ? a : -a;
            const S rx = cs * lx - sn * ly, ry = sn * lx + cs * ly;      // corner - centre, world frame
            const S gap = q[1] + ry;
#pragma unroll
            for (int gen = 0; gen < 2; ++gen) {
                const int row = 2 * c + gen;
                const T sg = gen == 0 ? mu : -mu;                       // e = n + sg t, n = (0,1), t = (1,0)
                phi[row] = gap;
                J[row][0] = S(sg);
                J[row][1] = S(T(1));
                J[row][2] = rx - ry * sg;                                // e . (z x r) = -e_x r_y + e_y r_x
                J[row][3] = S(T(0));
                J[row][4] = S(T(0));
            }
        }
        // rows 8..9: hand against the box
        irs_hand_box_rows<S, T>(q, sn, cs, a, rh, mu, J, phi, 8);
        // rows 10..11: hand against the ground
#pragma unroll
        for (int gen = 0; gen < 2; ++gen) {
            const int row = 10 + gen;
            const T sg = gen == 0 ? mu : -mu;
            phi[row] = q[4] - rh;
            J[row][0] = S(T(0)); J[row][1] = S(T(0)); J[row][2] = S(T(0));
            J[row][3] = S(sg);
            J[row][4] = S(T(1));
        }
        return iters;
    }
};

// Box pivoting with its step QP solved EXACTLY (12 rows), like PlanarHandExactModel.
struct BoxPivotExactModel : BoxPivotModel {
    static constexpr bool EXACT = true;
    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x_ext, const S* u, S* xn_ext) {
        irs_contact_step<BoxPivotExactModel, S>(p, x_ext, u, xn_ext);
    }
};

// examples/box_pushing/analysis/box_on_box.py:11-20 -- the reference's own 1-D statement of the scheme:
// a stiffness-controlled point (gain k) commanded to u pushes a mass m that sits in front of it.
//   x = [x_a, x_u], u = commanded x_a; params = {h, m, k, pgs_iters}
// One frictionless contact x_u - x_a >= 0, entered twice (the dual solver works on generator PAIRS).
// Exists to pin irs_contact_qp_step -- the code every contact functor shares -- against the closed
// form printed there: x+ = (m + h^2 k u) / (m + h^2 k) for both bodies once u > x_u.
struct BoxOnBoxModel {
    static constexpr int NX = 2, NU = 1, NPARAMS = 4;
    static constexpr int NC = 2;
    static constexpr bool HAS_JACOBIAN = false;
    IRS_HD static constexpr int perm(int k) { return k; }
    IRS_HD static constexpr int act(int) { return 0; }
    IRS_HD static int u_into_x(int) { return 0; }

    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x, const S* u, S* xn) {
        irs_contact_step<BoxOnBoxModel, S>(p, x, u, xn);
    }

    template <typename S>
    IRS_HD static int assemble(const ModelParams& p, const S* x, const S* u, S* q,
                               typename scalar_of<S>::type* Dinv, S* b, S (*J)[NX], S* phi) {
        using T = typename scalar_of<S>::type;
        const T h = T(p.v[0]), m = T(p.v[1]), k = T(p.v[2]);
        q[0] = x[0]; q[1] = x[1];
        Dinv[0] = T(1) / k; Dinv[1] = h * h / m;
        b[0] = k * (q[0] - u[0]);
        b[1] = q[1] * T(0);
#pragma unroll
        for (int r = 0; r < NC; ++r) {
            J[r][0] = S(T(-1)); J[r][1] = S(T(1));
            phi[r] = q[1] - q[0];
        }
        return (int)p.v[3];
    }
};

// examples/box_pushing (box_pushing_setup.py:6-19): the same box and disc seen from above -- no gravity,
// no ground, Kp = 500.  PINNED by the simulator data the reference ships,
// examples/box_pushing/analysis/{xu,dxdu}_quasistatic.npy (tests/golden/box_pushing_*.npy): the 80-step
// push and the simulator's Jacobians; they identify mass = 5, inertia = 1/6, half side 0.4995, r_hand = 0.1
// (the host passes them: irs_mpc_amd/systems.py).
//   x = [x_h, x_b, y_h, y_b, th_b], u = commanded hand position
//   params = {h, mass, inertia, half, mu, kp, r_hand, pgs_iters}
struct BoxPushModel {
    static constexpr int NX = 5, NU = 2, NPARAMS = 8;
    static constexpr int NC = 2;
    static constexpr bool HAS_JACOBIAN = false;
    IRS_HD static constexpr int perm(int k) { return BoxPivotModel::perm(k); }
    IRS_HD static constexpr int act(int j) { return 3 + j; }
    IRS_HD static int u_into_x(int j) { return perm(act(j)); }
    template <typename T>
    IRS_HD static T stiffness(const ModelParams& p, int) { return T(p.v[5]); }      // b[act(j)] = kp (q - u)

    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x_ext, const S* u, S* xn_ext) {
        irs_contact_step<BoxPushModel, S>(p, x_ext, u, xn_ext);
    }

    template <typename S>
    IRS_HD static int assemble(const ModelParams& p, const S* x_ext, const S* u, S* q,
                               typename scalar_of<S>::type* Dinv, S* b, S (*J)[NX], S* phi) {
        using T = typename scalar_of<S>::type;
        const T h = T(p.v[0]), mass = T(p.v[1]), inertia = T(p.v[2]), a = T(p.v[3]), mu = T(p.v[4]);
        const T kp = T(p.v[5]), rh = T(p.v[6]);
#pragma unroll
        for (int k = 0; k < NX; ++k) q[k] = x_ext[perm(k)];
        Dinv[0] = h * h / mass; Dinv[1] = h * h / mass; Dinv[2] = h * h / inertia;
        Dinv[3] = T(1) / kp; Dinv[4] = T(1) / kp;
        b[0] = S(T(0)); b[1] = S(T(0)); b[2] = S(T(0));
        b[3] = kp * (q[3] - u[0]); b[4] = kp * (q[4] - u[1]);
        S sn, cs;
        irs_sincos(q[2], sn, cs);
        irs_hand_box_rows<S, T>(q, sn, cs, a, rh, mu, J, phi, 0);
        return (int)p.v[7];
    }
};

// Box pushing with its step QP solved EXACTLY (2 rows), like PlanarHandExactModel.
struct BoxPushExactModel : BoxPushModel {
    static constexpr bool EXACT = true;
    template <typename S>
    IRS_HD static void step(const ModelParams& p, const S* x_ext, const S* u, S* xn_ext) {
        irs_contact_step<BoxPushExactModel, S>(p, x_ext, u, xn_ext);
    }
};

// ---- a step along a trajectory in TWO phases, for a caller that learns x_t before u_t ----------------------------
// The plant wave of the descent kernels (ctrlbox_mfma.hip, ctrlbox.hip) knows the realised state as soon as its
// previous step is done, but the control only when the solver wave has finished the tail -- and it idles in between.
// Everything of the step QP that depends on x alone is therefore prepared while the solver works: the contact
// geometry (J, phi), W = J D^-1 J' and its masked factorisation on the previous step's active set.  Once u arrives,
// the linear term, ONE pair of substitutions, the optimality test and the primal recovery remain (about a third of
// the step).  If the prepared set is not the optimal one (a contact made or broken), the step falls back to the full
// method, warm-started as before: the answer is the same either way.
// Only the actuated entries of b depend on u: b_a = (q_a - u_j) / Dinv_a (every contact model here: an impedance-
// controlled joint contributes K (q_a - u) with D_aa = K; irs_contact_qp_grad relies on the same structure).
template <class M>
struct irs_step_prepared {
    static constexpr int NX = M::NX, NC = M::NC;
    double q[NX], Dinv[NX], b0[NX], J[NC][NX], phi[NC], Mf[NC][NC], inv[NC];
    bool act[NC], ok;
};

template <class M>
IRS_HD void irs_step_along_prepare(const ModelParams& p, const double* x, unsigned warm, irs_step_prepared<M>& pre) {
    if constexpr (irs_contact_exact<M>::value) {
        constexpr int NX = M::NX, NC = M::NC, NU = M::NU;
        const double piv_rel = 1e-7;
        double u0[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) u0[j] = x[M::u_into_x(j)];          // b0 of the actuated rows = 0
        M::template assemble<double>(p, x, u0, pre.q, pre.Dinv, pre.b0, pre.J, pre.phi);
        pre.ok = warm != ~0u;
        double Wd[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            pre.act[i] = ((warm >> i) & 1u) != 0u && pre.ok;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double w = pre.J[i][0] * pre.Dinv[0] * pre.J[j][0];
#pragma unroll
                for (int k = 1; k < NX; ++k) w = w + pre.J[i][k] * pre.Dinv[k] * pre.J[j][k];
                pre.Mf[i][j] = w;
            }
            Wd[i] = pre.Mf[i][i];
        }
        // masked LDL' in row order (unit lower factor kept in the upper triangle), as in irs_contact_qp_dual_exact
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const double dj = pre.Mf[j][j];
            const bool ok = pre.act[j] && dj > piv_rel * Wd[j];
            pre.act[j] = ok;
            pre.inv[j] = ok ? irs_rcp_fast(dj) : 0.0;
#pragma unroll
            for (int i = j + 1; i < NC; ++i) pre.Mf[j][i] = pre.Mf[i][j] * pre.inv[j];
#pragma unroll
            for (int i = j + 1; i < NC; ++i)
#pragma unroll
                for (int k = j + 1; k <= i; ++k) pre.Mf[i][k] = pre.Mf[i][k] - pre.Mf[j][i] * pre.Mf[k][j];
        }
    }
}

template <class M>
IRS_HD void irs_step_along_finish(const ModelParams& p, const double* x, const double* u, const irs_step_prepared<M>& pre,
                                  double* xn, unsigned* warm) {
    if constexpr (irs_contact_exact<M>::value) {
        constexpr int NX = M::NX, NC = M::NC, NU = M::NU;
        bool good = pre.ok;
        double b[NX], lam[NC];
#pragma unroll
        for (int k = 0; k < NX; ++k) b[k] = pre.b0[k];
#pragma unroll
        for (int j = 0; j < NU; ++j) b[M::act(j)] = (pre.q[M::act(j)] - u[j]) / pre.Dinv[M::act(j)];
        if (pre.ok) {
            double r[NC], y[NC], Db[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) Db[k] = b[k] * pre.Dinv[k];
            double scale = 1e-30;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                double ri = pre.phi[i];
#pragma unroll
                for (int k = 0; k < NX; ++k) ri = ri - pre.J[i][k] * Db[k];
                r[i] = ri;
                scale = fmax(scale, fabs(ri));
            }
            const double tolv = 1e-10 * scale;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                double v = pre.act[j] ? -r[j] : 0.0;
#pragma unroll
                for (int k = 0; k < j; ++k) v = v - pre.Mf[k][j] * y[k];
                y[j] = v;
            }
#pragma unroll
            for (int j = NC - 1; j >= 0; --j) {
                double v = y[j] * pre.inv[j];
#pragma unroll
                for (int i = j + 1; i < NC; ++i) v = v - pre.Mf[j][i] * y[i];
                y[j] = v;
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                good = good && !(pre.act[i] && !(y[i] > 0.0));
                lam[i] = pre.act[i] ? y[i] : 0.0;
            }
            // slacks off the set, through J (W itself is not kept): s = r + J D^-1 J' lam
            double v[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                double f = 0.0;
#pragma unroll
                for (int i = 0; i < NC; ++i) f = f + pre.J[i][k] * lam[i];
                v[k] = f * pre.Dinv[k];
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                double sl = r[i];
#pragma unroll
                for (int k = 0; k < NX; ++k) sl = sl + pre.J[i][k] * v[k];
                good = good && (pre.act[i] || sl >= -tolv);
            }
        }
        if (irs_wave_all<double>(good)) {
            double qn[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                double f = -b[k];
#pragma unroll
                for (int i = 0; i < NC; ++i) f = f + pre.J[i][k] * lam[i];
                qn[k] = pre.q[k] + f * pre.Dinv[k];
            }
#pragma unroll
            for (int k = 0; k < NX; ++k) xn[M::perm(k)] = qn[k];
            unsigned mk = 0u;
#pragma unroll
            for (int i = 0; i < NC; ++i) mk |= pre.act[i] ? (1u << i) : 0u;
            *warm = mk;
        } else {
            // a contact made or broken: the full method, warm-started from the previous set as before -- on the
            // geometry that is already assembled
            double Wf[NC][NC], lamf[NC], qn[NX];
            irs_contact_qp_dual_exact<double, NX, NC>(pre.Dinv, b, pre.J, pre.phi, Wf, lamf, warm);
            irs_contact_qp_primal<double, NX, NC>(pre.q, pre.Dinv, b, pre.J, lamf, qn);
#pragma unroll
            for (int k = 0; k < NX; ++k) xn[M::perm(k)] = qn[k];
        }
    } else {
        M::template step<double>(p, x, u, xn);
    }
}

// One f64 step along a TRAJECTORY: models whose step QP is solved exactly take the previous step's active set
// as the starting guess of this one (`warm`: ~0u before the first step); every other model just steps.
template <class Model>
IRS_HD void irs_step_along(const ModelParams& p, const double* x, const double* u, double* xn, unsigned* warm) {
    if constexpr (irs_contact_exact<Model>::value) irs_contact_step<Model, double>(p, x, u, xn, warm);
    else Model::template step<double>(p, x, u, xn);
}
