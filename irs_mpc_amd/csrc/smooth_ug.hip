// Uniform-geometry sample pass: the u-only smoothing modes of the exact 8-row contact models
//   irs_lqr/quasistatic_dynamics.py:242-266  calc_B_zero_order    (ZERO_ORDER_B)
//   irs_lqr/quasistatic_dynamics.py:193-208  calc_AB_first_order  (FIRST_ORDER, u-only noise)
// -- the metric's workload (planar_hand, gradient modes "zero_order_B" / "first_order").
//
// In these modes the STATE of a timestep is not perturbed, only the command.  The step QP of sample s,
//     min_dq 1/2 dq'D dq + b(u_s)'dq   s.t.  phi + J dq >= 0,
// then has the SAME geometry (D, J, phi) for every sample of the timestep; only the actuated entries of b move,
// linearly in du.  Its dual  min_{lam >= 0} 1/2 lam'W lam + r'lam  has ONE Hessian W = J D^-1 J' per timestep and
// r = r0 + C du.  The general kernel (smooth.hip) carries J, W and a masked LDL' of W_AA in the registers of every
// lane (388 registers: one wave per SIMD) and re-factorises per sample.  Here
//   * everything uniform lives in LDS (C, J D^-1, the table) or in ONE set of registers per lane (W, r0: 52);
//   * the workgroup tabulates, once, for ALL 2^8 candidate active sets A the map  r -> [lam_A ; slacks off A]
//     (G_A: rows of -W_AA^-1 on A -- masked LDL' in row order with the same pivot rule as
//     irs_contact_qp_dual_exact, a dependent row drops out -- and of I + W M off A): 256 x 64 floats in LDS;
//   * a sample is solved by a few projected sweeps (a guess of A), then primal-dual active-set iterations that
//     cost ONE table row and ONE 8x8 product each: A <- (A minus rows with lam_i <= 0) + rows with slack < 0,
//     until the KKT conditions hold -- the exact optimum of the QP, certified per sample;
//   * samples that do not settle within the iteration cap are parked in the wave's LDS ring (as in smooth.hip)
//     and finished 64 at a time by the Goldfarb-Idnani dual active-set method (finite, no cycling), whose step
//     also costs one table row + one product;
//   * zero-order statistics: sum z z', sum z lam', sum z -- the primal step is a UNIFORM linear map of lam, so
//     it is applied once per workgroup to the sums (sum z df' = (sum z lam') (J D^-1) - ...), not per sample;
//   * first-order statistics: the derivative of the step through its active set depends on the active set
//     ALONE (geometry fixed), so the sample pass only counts active sets (integer LDS atomics: order-free,
//     deterministic) and B(I) is evaluated once per occupied set from the same table.
// Per-lane state: r, lam, a mask and the accumulators -- the kernel fits two (or more) waves per SIMD.
// The f64 nominal step f(x_t, u_t) is evaluated by the last wave of workgroup 0 of every timestep, cooperatively
// (lane c owns contact row c; an 8x8 masked LDL' spread over the 64 lanes), from the f32 pipeline's active set,
// re-checked against the KKT conditions in f64.
#include "smooth_common.hpp"

namespace {

#ifndef IRS_UG_SWEEPS
#define IRS_UG_SWEEPS 6
#endif
#ifndef IRS_UG_PDAS
#define IRS_UG_PDAS 4
#endif
#ifndef IRS_UG_BLOCK
#define IRS_UG_BLOCK 512
#endif
constexpr int kUgBlock = IRS_UG_BLOCK;
constexpr int kUgSweeps = IRS_UG_SWEEPS;      // projected sweeps that guess the active set
constexpr int kUgPdas = IRS_UG_PDAS;          // primal-dual active-set iterations of the first attempt
constexpr int kUgTabStride = 76;              // floats per table entry: 64 (G', column-major) + 8 (tau: 1 off the reduced
                                              // set, 0 on it) + the reduced set + pad; 304 B: 16-byte aligned rows
#ifndef IRS_UG_NOM_ROUNDS
#define IRS_UG_NOM_ROUNDS 2
#endif
constexpr int kUgNomRounds = IRS_UG_NOM_ROUNDS;   // sample rounds the nominal wave sits out (its f64 step takes about that long)
constexpr int kUgRing = 128;                  // parked samples per wave: < 64 waiting + <= 64 new ones

template <class Model>
struct alignas(16) UgLds {
    static constexpr int NC = Model::NC, n = Model::NX, m = Model::NU;
    float tab[1 << NC][kUgTabStride];     // first: every ds_read_b128 of the sample loop is 16-byte aligned
    float W[NC][NC];          // dual Hessian J D^-1 J'
    float invw[NC];           // omega / W_ii
    float Wd[NC];             // W_ii
    float r0[NC];             // phi - J D^-1 b(u_t)
    float phi[NC];
    float C[m][NC];           // r = r0 + sum_j C[j][:] du_j
    float JD[n][NC];          // JD[k][c] = J[c][k] D^-1_k   (primal recovery: dq_k = sum_c JD[k][c] lam_c - Db_k)
    float Jc[NC][n];          // J
    float Dinv[n];
    float Db0[n];             // D^-1 b(u_t)
    float DK[m];              // D^-1_act(j) K_j  (= 1 up to rounding)
    float q[n];
};

__device__ __forceinline__ float ug_uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// the uniform operands the sample loop keeps in registers
template <int NC>
struct UgUni {
    float W[NC * (NC + 1) / 2];
    float invw[NC];
    float r0[NC];
    __device__ __forceinline__ float w(int i, int j) const {
        return i <= j ? W[i * NC - i * (i - 1) / 2 + (j - i)] : W[j * NC - j * (j - 1) / 2 + (i - j)];
    }
};

// v = G_a x + tol tau_a (table entry a, G stored column-major: row c of the entry = column c of G_a; tau_i = 1 for a
// row off the reduced set -- whose v_i is a slack, optimal when >= -tol -- and 0 on it -- v_i a multiplier, optimal
// when > 0: with the tolerance folded in, "row i is optimal" is v_i > 0 for both kinds).  All 18 reads are issued
// before the first use: a wave has one partner on its SIMD to hide the LDS latency behind.
template <class Model>
__device__ __forceinline__ unsigned ug_lookup(const UgLds<Model>& S, unsigned a, const float* x, float tol, float* v,
                                              float* tau) {
    constexpr int NC = Model::NC;
    static_assert(NC == 8, "table entries are 8 x 8");
    const float4* e = reinterpret_cast<const float4*>(&S.tab[a][0]);
    float4 g4[2 * NC + 2];
#pragma unroll
    for (int c = 0; c < 2 * NC + 2; ++c) g4[c] = e[c];
    const unsigned ap = __float_as_uint(S.tab[a][72]);
    tau[0] = g4[16].x; tau[1] = g4[16].y; tau[2] = g4[16].z; tau[3] = g4[16].w;
    tau[4] = g4[17].x; tau[5] = g4[17].y; tau[6] = g4[17].z; tau[7] = g4[17].w;
#pragma unroll
    for (int i = 0; i < NC; ++i) v[i] = tau[i] * tol;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float4 lo = g4[2 * c], hi = g4[2 * c + 1];
        v[0] = fmaf(lo.x, x[c], v[0]); v[1] = fmaf(lo.y, x[c], v[1]);
        v[2] = fmaf(lo.z, x[c], v[2]); v[3] = fmaf(lo.w, x[c], v[3]);
        v[4] = fmaf(hi.x, x[c], v[4]); v[5] = fmaf(hi.y, x[c], v[5]);
        v[6] = fmaf(hi.z, x[c], v[6]); v[7] = fmaf(hi.w, x[c], v[7]);
    }
    return ap;
}

// r = r0 + C du and the tolerance of the optimality tests (that of irs_contact_qp_dual_exact)
template <class Model>
__device__ __forceinline__ float ug_rhs(const UgUni<Model::NC>& U, const UgLds<Model>& S, const float* du, float* r) {
    constexpr int NC = Model::NC, m = Model::NU;
#pragma unroll
    for (int c = 0; c < NC; ++c) r[c] = U.r0[c];
#pragma unroll
    for (int j = 0; j < m; ++j) {
        const float4 lo = *reinterpret_cast<const float4*>(&S.C[j][0]);
        const float4 hi = *reinterpret_cast<const float4*>(&S.C[j][4]);
        r[0] = fmaf(lo.x, du[j], r[0]); r[1] = fmaf(lo.y, du[j], r[1]);
        r[2] = fmaf(lo.z, du[j], r[2]); r[3] = fmaf(lo.w, du[j], r[3]);
        r[4] = fmaf(hi.x, du[j], r[4]); r[5] = fmaf(hi.y, du[j], r[5]);
        r[6] = fmaf(hi.z, du[j], r[6]); r[7] = fmaf(hi.w, du[j], r[7]);
    }
    float scale = 1e-30f;
#pragma unroll
    for (int c = 0; c < NC; ++c) scale = fmaxf(scale, fabsf(r[c]));
    return 1e-6f * scale;
}

// FIRST ATTEMPT: sweeps guess the active set, primal-dual active-set iterations settle it.  true: lam is the exact
// optimum (KKT certified); false: `a` is the set the iteration stopped on (a warm start for ug_full).
template <class Model>
__device__ __forceinline__ bool ug_try(const UgUni<Model::NC>& U, const UgLds<Model>& S, const float* r, float tolv,
                                       float* lam, unsigned& a_out, int diag = 0) {
    constexpr int NC = Model::NC;
    float g[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) { lam[i] = 0.f; g[i] = r[i]; }
#pragma unroll 2
    for (int sw = 0; sw < ((diag & 64) ? 0 : kUgSweeps); ++sw) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const float nw = fmaxf(fmaf(-g[i], U.invw[i], lam[i]), 0.f);
            const float dl = nw - lam[i];
            lam[i] = nw;
#pragma unroll
            for (int j = 0; j < NC; ++j) g[j] = fmaf(U.w(j, i), dl, g[j]);
        }
    }
    unsigned a = 0u;
#pragma unroll
    for (int i = 0; i < NC; ++i) a |= (lam[i] > 0.f) ? (1u << i) : 0u;
    bool done = (diag & 32) != 0;
    for (int it = 0; it < kUgPdas; ++it) {
        float v[NC], tau[NC];
        const unsigned ap = ug_lookup<Model>(S, a, r, tolv, v, tau);
        // rows that are not optimal flip: a multiplier <= 0 leaves the set, a slack < -tol joins it
        // (the sign bits, shifted in one v_alignbit each: row 0 ends in bit 0.  A multiplier of exactly +0 stays: it
        // satisfies the KKT conditions as it is)
        unsigned bad = 0u;
#pragma unroll
        for (int i = NC - 1; i >= 0; --i) bad = __builtin_amdgcn_alignbit(bad, __float_as_uint(v[i]), 31);
        if (!done) {
            if (bad == 0u) {
#pragma unroll
                for (int i = 0; i < NC; ++i) lam[i] = fmaf(-tau[i], v[i], v[i]);      // v on the set, 0 off it
                done = true;
            }
            a = ap ^ bad;
        }
        if (__all(done)) break;
    }
    a_out = a;
    return done;
}

// FULL METHOD from a warm set: the Goldfarb-Idnani dual active-set method in the dual variables, restated from
// irs_contact_qp_dual_exact (contact_models.hpp) with every factorisation replaced by a table row:
//   repair rounds: lam_A = -W_AA^-1 r_A; rows with lam_i <= 0 leave (<= 3 rounds; else start from lam = 0);
//   loop: p = most violated row off A; u = G_A W[:,p] gives rho = -u on A and the slack rates s = u off A in one
//   product; step to the first of { slack_p = 0 (p joins), some lam_i = 0 (i leaves) }.  Finite, no cycling.
template <class Model>
__device__ __forceinline__ void ug_full(const UgLds<Model>& S, const float* r, float tolv, unsigned a, float* lam,
                                        int cap = 4 * Model::NC) {
    constexpr int NC = Model::NC;
    constexpr float kBig = 3.0e38f, piv_rel = 1e-5f;
    float g[NC];
    bool valid = false;
    for (int round = 0; round < 3; ++round) {
        if (__all(valid)) break;
        float v[NC], tau[NC];
        const unsigned ap = ug_lookup<Model>(S, a, r, 0.f, v, tau);
        if (!valid) {
            unsigned negb = 0u;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const bool in = tau[i] == 0.f;
                negb |= (in && !(v[i] > 0.f)) ? (1u << i) : 0u;
                lam[i] = in ? v[i] : 0.f;
                g[i] = in ? 0.f : v[i];
            }
            valid = negb == 0u;
            a = ap & ~negb;
        }
    }
    if (!valid) {
        a = 0u;
#pragma unroll
        for (int i = 0; i < NC; ++i) { lam[i] = 0.f; g[i] = r[i]; }
    }
    int p = -1;
    bool done = false;
    for (int it = 0; it < cap; ++it) {
        if (!done && p < 0) {
            float vmin = kBig;
            int c = 0;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const float vv = ((a >> i) & 1u) ? kBig : g[i];
                if (vv < vmin) { vmin = vv; c = i; }
            }
            if (vmin >= -tolv) done = true;
            else p = c;
        }
        if (__all(done)) break;
        const int pp = (done || p < 0) ? 0 : p;
        float wp[NC], u[NC], tau[NC];
        {
            const float4 lo = *reinterpret_cast<const float4*>(&S.W[pp][0]);
            const float4 hi = *reinterpret_cast<const float4*>(&S.W[pp][4]);
            wp[0] = lo.x; wp[1] = lo.y; wp[2] = lo.z; wp[3] = lo.w;
            wp[4] = hi.x; wp[5] = hi.y; wp[6] = hi.z; wp[7] = hi.w;
        }
        ug_lookup<Model>(S, a, wp, 0.f, u, tau);
        // in arithmetic rather than selects (tau_i = 1 off the reduced set, 0 on it): rho = -(1 - tau) u on the set,
        // the slack rates s = tau u off it -- a third of the instructions of the select form (400 -> ~270 per step)
        const float wpp = S.Wd[pp];
        float gp = 0.f, zp = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const bool isp = i == pp;
            gp = isp ? g[i] : gp;
            zp = isp ? u[i] : zp;
        }
        const bool full_ok = zp > piv_rel * wpp;
        const float t2 = full_ok ? -gp * irs_rcp_fast(zp) : kBig;
        float rho[NC], qv[NC];
        float t1 = kBig;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            rho[i] = fmaf(tau[i], u[i], -u[i]);
            qv[i] = rho[i] > 0.f ? lam[i] * irs_rcp_fast(rho[i]) : kBig;
            t1 = fminf(t1, qv[i]);
        }
        int kb = 0;                                        // the FIRST row that attains the minimum
#pragma unroll
        for (int i = NC - 1; i >= 0; --i) kb = (qv[i] == t1) ? i : kb;
        const float tmin = fminf(t1, t2);
        if (!done && !(tmin < kBig)) done = true;          // no step possible: infeasible primal, keep lam
        const float tt = done ? 0.f : tmin;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            g[i] = fmaf(tt * tau[i], u[i], g[i]);
            lam[i] = fmaxf(fmaf(-tt, rho[i], lam[i]) + ((i == pp) ? tt : 0.f), 0.f);
        }
        if (!done) {
            const bool full = t2 <= t1;
            if (full) {
                a |= 1u << pp;
                p = -1;
            } else {
                a &= ~(1u << kb);
#pragma unroll
                for (int i = 0; i < NC; ++i) lam[i] = (i == kb) ? 0.f : lam[i];
            }
        }
    }
}

// One table entry: G_a for the candidate set `a`, by the masked LDL' in row order of irs_contact_qp_dual_exact_try
// (pivot rule included: a dependent row leaves the set).  `half` (0/1): the four columns this lane writes.
template <class Model>
__device__ __forceinline__ void ug_build_entry(UgLds<Model>& S, unsigned a, int half) {
    constexpr int NC = Model::NC;
    constexpr float piv_rel = 1e-5f;
    float M_[NC][NC], inv[NC], W[NC][NC];
    bool act[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        act[i] = ((a >> i) & 1u) != 0u;
#pragma unroll
        for (int j = 0; j < NC; ++j) W[i][j] = S.W[i][j];
#pragma unroll
        for (int j = 0; j <= i; ++j) M_[i][j] = W[i][j];
    }
    unsigned ap = 0u;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const float dj = M_[j][j];
        const bool ok = act[j] && dj > piv_rel * W[j][j];
        act[j] = ok;
        ap |= ok ? (1u << j) : 0u;
        inv[j] = ok ? irs_rcp_fast(dj) : 0.f;
#pragma unroll
        for (int i = j + 1; i < NC; ++i) M_[j][i] = M_[i][j] * inv[j];
#pragma unroll
        for (int i = j + 1; i < NC; ++i)
#pragma unroll
            for (int k = j + 1; k <= i; ++k) M_[i][k] = M_[i][k] - M_[j][i] * M_[k][j];
    }
    float* e = &S.tab[a][0];
    for (int cc = 0; cc < NC / 2; ++cc) {
        const int c = half * (NC / 2) + cc;
        // y = -W_AA^-1 e_c on the set (zero off it)
        float y[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            float v = (act[j] && j == c) ? -1.f : 0.f;
#pragma unroll
            for (int k = 0; k < j; ++k) v = v - M_[k][j] * y[k];
            y[j] = v;
        }
#pragma unroll
        for (int j = NC - 1; j >= 0; --j) {
            float v = y[j] * inv[j];
#pragma unroll
            for (int i = j + 1; i < NC; ++i) v = v - M_[j][i] * y[i];
            y[j] = v;
        }
        float col[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            float s = (i == c) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < NC; ++j) s = fmaf(W[i][j], y[j], s);
            col[i] = act[i] ? y[i] : s;
        }
        *reinterpret_cast<float4*>(e + 8 * c) = make_float4(col[0], col[1], col[2], col[3]);
        *reinterpret_cast<float4*>(e + 8 * c + 4) = make_float4(col[4], col[5], col[6], col[7]);
    }
    if (half == 0) {
        *reinterpret_cast<float4*>(e + 64) = make_float4(act[0] ? 0.f : 1.f, act[1] ? 0.f : 1.f, act[2] ? 0.f : 1.f, act[3] ? 0.f : 1.f);
        *reinterpret_cast<float4*>(e + 68) = make_float4(act[4] ? 0.f : 1.f, act[5] ? 0.f : 1.f, act[6] ? 0.f : 1.f, act[7] ? 0.f : 1.f);
        e[72] = __uint_as_float(ap);
    }
}

// ---- the f64 nominal step, one wave, cooperatively ------------------------------------------------------------
// lane c < NC owns contact row c of the f64 geometry; lane (i,k) = 8 i + k owns entry (i,k) of the dual Hessian and
// of its masked LDL' (row order, pivot rule of the f64 solver: 1e-7).  The active set comes from the f32 pipeline at
// du = 0 and is re-checked -- and, if need be, corrected by primal-dual steps -- against the KKT conditions in f64
// (tolerance 1e-10 of max |r|, as irs_contact_qp_dual_exact<double>).  The primal solution of the step QP is unique,
// so whatever route finds a KKT point finds f(x_t, u_t).  Not settled within 8 corrections (never observed): the f32
// solution stands (1e-6 accurate).
struct UgNomLds {
    double J[8][7];
    double q[7], Dinv[7], b[7];
};

// (a) the f64 geometry: independent of the f32 prologue and of the table, so the nominal wave runs it BEFORE the
// workgroup's first barrier, beside wave 0's f32 assembly.  Leaves J, q, D^-1, b in LDS and returns r_c on lane c.
template <class Model>
__device__ __forceinline__ double ug_nominal_geometry(const SmoothArgs& a, UgNomLds& L, int t, int lane) {
    constexpr int NC = Model::NC, n = Model::NX, m = Model::NU;
    double x64[n], u64[m], q[n], Dinv[n], b[n], J[NC][n], phi[NC];
#pragma unroll
    for (int i = 0; i < n; ++i) x64[i] = a.x_trj[(size_t)t * n + i];
#pragma unroll
    for (int j = 0; j < m; ++j) u64[j] = a.u_trj[(size_t)t * m + j];
    Model::template assemble<double>(a.p, x64, u64, q, Dinv, b, J, phi);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int k = 0; k < n; ++k) L.J[c][k] = J[c][k];
#pragma unroll
        for (int k = 0; k < n; ++k) { L.q[k] = q[k]; L.Dinv[k] = Dinv[k]; L.b[k] = b[k]; }
    }
    // r_c on lane c (phi is the only per-row quantity not in LDS)
    double rr = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        double s = phi[c];
#pragma unroll
        for (int k = 0; k < n; ++k) s -= J[c][k] * (b[k] * Dinv[k]);
        rr = (lane == c) ? s : rr;
    }
    wave_sync();
    return rr;
}

// (b) after the table exists: the f32 pipeline's active set at du = 0, then the cooperative f64 solve and its check
template <class Model>
__device__ __forceinline__ void ug_nominal(const SmoothArgs& a, const UgUni<Model::NC>& U, const UgLds<Model>& S,
                                           UgNomLds& L, double rr, int t, int lane) {
    constexpr int NC = Model::NC, n = Model::NX, m = Model::NU;
    static_assert(NC == 8 && n <= 8, "lane (i,k) = 8 i + k owns entry (i,k) of the 8 x 8 dual Hessian");
    // (1) the f32 pipeline at du = 0: the active set (every lane computes the same)
    float du0[m], r32[NC], lam32[NC];
#pragma unroll
    for (int j = 0; j < m; ++j) du0[j] = 0.f;
    const float tolv = ug_rhs<Model>(U, S, du0, r32);
    unsigned am = 0u;
    if (!ug_try<Model>(U, S, r32, tolv, lam32, am)) ug_full<Model>(S, r32, tolv, am, lam32);
    unsigned set = 0u;
#pragma unroll
    for (int i = 0; i < NC; ++i) set |= (lam32[i] > 0.f) ? (1u << i) : 0u;
    set = (unsigned)__builtin_amdgcn_readfirstlane((int)set);
    {
        // entry (i,k) of W on lane 8 i + k
        const int li = lane >> 3, lk = lane & 7;
        double w = 0.0;
#pragma unroll
        for (int kk = 0; kk < n; ++kk) w += L.J[li][kk] * L.Dinv[kk] * L.J[lk][kk];
        double scale = 1e-300;
#pragma unroll
        for (int c = 0; c < NC; ++c) scale = fmax(scale, fabs(__shfl(rr, c, 64)));
        const double tol64 = 1e-10 * scale;
        double lamd = 0.0;          // lane i < NC: lam_i
        bool settled = false;
        for (int corr = 0; corr < 8 && !settled; ++corr) {
            // masked LDL' of W on `set`, entry-parallel
            double M = w, Lf = 0.0;
            unsigned eff = 0u;
            double invd[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double dj = __shfl(M, 9 * j, 64), wjj = __shfl(w, 9 * j, 64);
                const bool ok = ((set >> j) & 1u) && dj > 1e-7 * wjj;
                eff |= ok ? (1u << j) : 0u;
                invd[j] = ok ? 1.0 / dj : 0.0;
                const double Mij = __shfl(M, 8 * li + j, 64), Mkj = __shfl(M, 8 * lk + j, 64);
                if (lk == j && li > j) Lf = Mij * invd[j];
                if (li > j && lk > j) M -= Mij * invd[j] * Mkj;
            }
            // forward / diagonal / backward substitution on lanes 0..7
            const bool mine = lane < NC && ((eff >> lane) & 1u);
            double y = mine ? -rr : 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double yj = __shfl(y, j, 64);
                const double Lij = __shfl(Lf, 8 * (lane & 7) + j, 64);
                if (lane < NC && lane > j) y -= Lij * yj;
            }
            double invme = 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) invme = (lane == j) ? invd[j] : invme;
            y *= invme;
#pragma unroll
            for (int j = NC - 1; j >= 0; --j) {
                const double yj = __shfl(y, j, 64);
                const double Lji = __shfl(Lf, 8 * j + (lane & 7), 64);
                if (lane < j) y -= Lji * yj;
            }
            lamd = mine ? y : 0.0;
            // slacks: s_i = r_i + sum_k W_ik lam_k, reduced over the 8 lanes of row i
            double pr = w * __shfl(lamd, lk, 64);
            pr += __shfl_xor(pr, 1, 64);
            pr += __shfl_xor(pr, 2, 64);
            pr += __shfl_xor(pr, 4, 64);
            const double sl = rr + __shfl(pr, 8 * (lane & 7), 64);       // lane i < NC
            const bool neg = lane < NC && mine && !(lamd > 0.0);
            const bool viol = lane < NC && !mine && sl < -tol64;
            const unsigned negb = (unsigned)(__ballot(neg) & 0xffull), violb = (unsigned)(__ballot(viol) & 0xffull);
            if ((negb | violb) == 0u) settled = true;
            else set = (eff & ~negb) | violb;
        }
        // f_k = q_k + D^-1_k (sum_c J[c][k] lam_c - b_k) on lane k < n
        double f = 0.0;
        if (settled) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) acc += L.J[c][lane < n ? lane : 0] * __shfl(lamd, c, 64);
            if (lane < n) f = L.q[lane] + L.Dinv[lane] * (acc - L.b[lane]);
        } else if (lane < n) {
            float acc = -S.Db0[lane];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc = fmaf(S.JD[lane][c], lam32[c], acc);
            f = (double)(S.q[lane] + acc);
        }
        if (lane < n)
            __hip_atomic_store(a.fnom + (size_t)t * n + Model::perm(lane), f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// -DIRS_UG_STAMPS (tuning builds only): s_memtime at the phase boundaries of every workgroup -> a device buffer that
// irs_debug_ug_stamps copies out (tests/tools/ug_stamps.py)
#ifdef IRS_UG_STAMPS
constexpr int kUgStampSlots = 12, kUgStampWgs = 2048;
__device__ unsigned long long ug_stamps[kUgStampWgs * kUgStampSlots];
#define UG_STAMP(slot)                                                                          \
    do {                                                                                        \
        if ((threadIdx.x & 63) == 0) {                                                          \
            const int wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                \
            if (wg_ < kUgStampWgs && ((slot) < 8 ? threadIdx.x == 0 : true))                    \
                ug_stamps[wg_ * kUgStampSlots + (slot)] = __builtin_amdgcn_s_memtime();         \
        }                                                                                       \
    } while (0)
#else
#define UG_STAMP(slot) do {} while (0)
#endif

template <class Model, int MODE, bool RNG, bool FUSE>
__global__ __launch_bounds__(kUgBlock) void smooth_ug_kernel(SmoothArgs a) {
    using TR = SmoothTraits<Model, MODE>;
    constexpr int n = TR::n, m = TR::m, P = TR::P, NC = Model::NC;
    constexpr int BLOCK = kUgBlock, NW = BLOCK / 64;
    static_assert(TR::Z0 == n && NC == 8 && m == 4, "u-only smoothing of an 8-row contact model with 4 commands");
    static_assert(P < kBlock, "the hand-off's group reduction");
    // internal statistics of the zero-order pass: [upper Gram of du (NG) | du lam' (m x NC) | sum du (m)]
    constexpr int NG = m * (m + 1) / 2;
    constexpr int PI = TR::FIRST_B ? 1 : NG + m * NC + m;
    constexpr int PPI = irs_reduce_pad(TR::FIRST_B ? n * m : PI);
    constexpr int QE = m + 1;
    __shared__ UgLds<Model> S;
    __shared__ UgNomLds nomL;
    __shared__ float ring_all[NW * kUgRing * QE];
    __shared__ float red[NW * (PPI > TR::PP ? PPI : TR::PP)];
    __shared__ float toti[PPI];
    __shared__ unsigned hist[TR::FIRST_B ? (1 << NC) : 1];
    __shared__ double red64[TR::NGRP * P];
    __shared__ double tot[P];
    __shared__ FinalizeLds<Model, MODE> fin;
    __shared__ int s_ticket;

    const int t = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    UG_STAMP(0);
    // ---- prologue: the timestep's geometry (wave 0; the nominal wave its f64 twin), then the table (everyone) ------
    const bool nominal_wave = blk == 0 && wave == NW - 1;
    double nom_r = 0.0;
    if (nominal_wave && !(a.diag & 2)) nom_r = ug_nominal_geometry<Model>(a, nomL, t, lane);
    if (wave == 0) {
        // the model's QP assembly (trigonometry, closest points): one wave, every lane the same; lane 0 stores
        float xb[n], ub[m], q[n], Dinv[n], b0[n], J[NC][n], phi[NC];
#pragma unroll
        for (int i = 0; i < n; ++i) xb[i] = (float)a.x_trj[(size_t)t * n + i];
#pragma unroll
        for (int j = 0; j < m; ++j) ub[j] = (float)a.u_trj[(size_t)t * m + j];
        Model::template assemble<float>(a.p, xb, ub, q, Dinv, b0, J, phi);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                S.phi[i] = phi[i];
#pragma unroll
                for (int k = 0; k < n; ++k) S.Jc[i][k] = J[i][k];
            }
#pragma unroll
            for (int k = 0; k < n; ++k) { S.Db0[k] = b0[k] * Dinv[k]; S.q[k] = q[k]; S.Dinv[k] = Dinv[k]; }
#pragma unroll
            for (int j = 0; j < m; ++j) S.DK[j] = Dinv[Model::act(j)] * Model::template stiffness<float>(a.p, j);
        }
    }
    if constexpr (TR::FIRST_B) {
        for (int q = tid; q < (1 << NC); q += BLOCK) hist[q] = 0u;
    }
    __syncthreads();
    // what follows from it, one quantity per thread (the expressions of irs_contact_qp_dual_exact_try):
    //   W = J D^-1 J' (entry (i,j) from the row pair (max, min): symmetric to the bit), r0 = phi - J D^-1 b,
    //   J D^-1, C = J D^-1[:, act] K
    if (tid < NC * NC) {
        const int i = tid / NC, j = tid % NC, hi = i > j ? i : j, lo = i > j ? j : i;
        float w = (S.Jc[hi][0] * S.Dinv[0]) * S.Jc[lo][0];
#pragma unroll
        for (int k = 1; k < n; ++k) w = w + (S.Jc[hi][k] * S.Dinv[k]) * S.Jc[lo][k];
        S.W[i][j] = w;
        if (i == j) {
            S.Wd[i] = w;
            S.invw[i] = (float)kContactPgsOmega * irs_rcp_fast(w);
        }
    } else if (tid < NC * NC + NC) {
        const int i = tid - NC * NC;
        float ri = S.phi[i];
#pragma unroll
        for (int k = 0; k < n; ++k) ri = ri - S.Jc[i][k] * S.Db0[k];
        S.r0[i] = ri;
    } else if (tid < 2 * NC * NC + NC) {
        const int e = tid - (NC * NC + NC), k = e / NC, c = e % NC;
        if (k < n) S.JD[k][c] = S.Jc[c][k] * S.Dinv[k];
    } else if (tid < 2 * NC * NC + NC + m * NC) {
        const int e = tid - (2 * NC * NC + NC), j = e / NC, c = e % NC;
        S.C[j][c] = (S.Jc[c][Model::act(j)] * S.Dinv[Model::act(j)]) * Model::template stiffness<float>(a.p, j);
    }
    __syncthreads();
    UG_STAMP(1);
    if (!(a.diag & 4))
        for (int e = tid; e < 2 * (1 << NC); e += BLOCK) ug_build_entry<Model>(S, (unsigned)(e >> 1), e & 1);
    __syncthreads();

    UG_STAMP(2);
    // the uniform operands of the sweeps, in VECTOR registers (one copy per lane).  Measured on gfx950
    // (tools/microbench/valu_issue.hip): a v_fmac_f32 with ONE scalar-register operand costs a SIMD 6.4 cycles at two
    // waves per SIMD (12.4 at one) against 2.9 with vector operands only -- scalar operands are read through a path
    // the four SIMDs of a CU share -- so keeping W in SGPRs would more than halve the rate of the sweeps.
    UgUni<NC> U;
    {
        int q = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int j = i; j < NC; ++j) U.W[q++] = S.W[i][j];
#pragma unroll
        for (int i = 0; i < NC; ++i) { U.invw[i] = S.invw[i]; U.r0[i] = S.r0[i]; }
    }

    float acc[PPI];
#pragma unroll
    for (int i = 0; i < PPI; ++i) acc[i] = 0.f;

    if (nominal_wave && !(a.diag & 2)) ug_nominal<Model>(a, U, S, nomL, nom_r, t, lane);
    if (nominal_wave) UG_STAMP(8);
    if (!(a.diag & 8)) {
        // 64-sample blocks are dealt round robin to the waves of the timestep: rounds 0 .. kUgNomRounds - 1 to all but
        // the nominal wave (which is busy with the f64 step for about that long), later rounds to all of them
        const int V = a.nblk * NW;
        const int me = nominal_wave ? V - 1 : (blk == 0 ? wave : blk * NW - 1 + wave);
        const int nblocks = (a.N + 63) / 64;
        int round = nominal_wave ? kUgNomRounds : 0;
        auto block_of = [&](int k) { return k < kUgNomRounds ? k * (V - 1) + me : kUgNomRounds * (V - 1) + (k - kUgNomRounds) * V + me; };
        float* ring = ring_all + wave * (kUgRing * QE);
        int qhead = 0, qtail = 0;                           // wave-uniform
        int bk = block_of(round);
        // the perturbations of the NEXT fresh trip are requested while this one computes: a trip is ~1.5 us of
        // arithmetic, an HBM round trip ~0.5-1 us, and a SIMD holds two waves -- nothing else would hide it.  (No
        // other vector-memory operation sits inside a trip, so the wait lands where the sample is consumed.)
        float du_next[m];
#pragma unroll
        for (int j = 0; j < m; ++j) du_next[j] = 0.f;
        auto request = [&](int block) {
            if constexpr (!RNG) {
                if (block < nblocks) {
                    const int sidx = block * 64 + lane;
                    const size_t row = (size_t)t * a.N + (sidx < a.N ? sidx : a.N - 1);
                    load_row<m>(a.du + row * m, du_next);
                }
            }
        };
        request(bk);
        while (true) {
            const bool fresh = bk < nblocks;
            const int pending = qtail - qhead;
            const bool flush = pending >= 64 || (!fresh && pending > 0);
            if (!fresh && !flush) break;
            float du[m], r[NC], lam[NC];
            bool on, fin_ = true;
            if (flush) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int take = min(64, pending);
                on = lane < take;
                const int slot = (qhead + (on ? lane : 0)) & (kUgRing - 1);
#pragma unroll
                for (int j = 0; j < m; ++j) du[j] = ring[slot * QE + j];
                const unsigned wm = __float_as_uint(ring[slot * QE + m]);
                const float tolv = ug_rhs<Model>(U, S, du, r);
                ug_full<Model>(S, r, tolv, wm, lam, (a.diag & 128) ? 8 : ((a.diag & 256) ? 12 : 4 * NC));
                qhead += take;
            } else {
                const int sidx = bk * 64 + lane;
                on = sidx < a.N;
                if constexpr (RNG) {
                    const unsigned long long gidx = a.sample_offset + (unsigned long long)sidx;
#pragma unroll
                    for (int j = n / 4; j < (n + m + 3) / 4; ++j) {
                        float g4[4];
                        philox_normal4(gidx, (unsigned)t, (unsigned)j, a.iter, a.seed, g4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int idx = 4 * j + c;
                            if (idx >= n && idx < n + m) du[idx - n] = __fmul_rn(g4[c], a.std[idx]);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < m; ++j) du[j] = du_next[j];
                    request(block_of(round + 1));
                }
                const float tolv = ug_rhs<Model>(U, S, du, r);
                unsigned mask = 0u;
                fin_ = ug_try<Model>(U, S, r, tolv, lam, mask, a.diag);
                if (a.diag & 16) fin_ = true;
                const bool hard = on && !fin_;
                const unsigned long long bal = __ballot(hard);
                if (hard) {
                    const int slot = (qtail + __popcll(bal & ((1ull << lane) - 1ull))) & (kUgRing - 1);
#pragma unroll
                    for (int j = 0; j < m; ++j) ring[slot * QE + j] = du[j];
                    ring[slot * QE + m] = __uint_as_float(mask);
                }
                qtail += __popcll(bal);
                bk = block_of(++round);
            }
            const bool use = on && fin_;
            // a non-finite perturbation must not vanish in a clamp: poison the statistics (tested on the bit pattern)
            bool nonfinite = false;
#pragma unroll
            for (int j = 0; j < m; ++j) nonfinite = nonfinite || irs_nonfinite_bits(du[j]);
            if constexpr (TR::FIRST_B) {
                unsigned I = 0u;
#pragma unroll
                for (int i = 0; i < NC; ++i) I |= (lam[i] * S.Wd[i] > (float)kContactActiveTol) ? (1u << i) : 0u;
                if (use) __hip_atomic_fetch_add(&hist[I], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (on && nonfinite) acc[0] = irs_poison();
            } else {
                float zz[m];
#pragma unroll
                for (int j = 0; j < m; ++j) zz[j] = use ? du[j] : 0.f;
                int q = 0;
#pragma unroll
                for (int i = 0; i < m; ++i)
#pragma unroll
                    for (int j = i; j < m; ++j) { acc[q] = fmaf(zz[i], zz[j], acc[q]); ++q; }
#pragma unroll
                for (int i = 0; i < m; ++i)
#pragma unroll
                    for (int c = 0; c < NC; ++c) { acc[q] = fmaf(zz[i], lam[c], acc[q]); ++q; }
#pragma unroll
                for (int i = 0; i < m; ++i) { acc[q] += zz[i]; ++q; }
                if (on && nonfinite) acc[0] = irs_poison();
            }
        }
    }

    UG_STAMP(3);
    if (blk == 0 && wave == 1) UG_STAMP(9);
    // ---- workgroup statistics in the layout of `sums` (include/irs_hip.h) --------------------------------------
    if constexpr (TR::FIRST_B) {
        // sum over the samples of B(I_s) = sum over the occupied active sets of count(I) B(I), with
        //   Y = W_II^+ J[:, act] = -M_I J[:, act],   B[k][c] = [k = act c] - D^-1_k sum_i J[i][k] Y[i][c]
        // (irs_contact_qp_grad, WITH_A = false; the table's pivot rule is the derivative's, 1e-5)
        __syncthreads();
        unsigned poison_bits = __float_as_uint(acc[0]);
        asm volatile("" : "+v"(poison_bits));
#pragma unroll
        for (int i = 0; i < PPI; ++i) acc[i] = 0.f;
        for (int I = tid; I < (1 << NC); I += BLOCK) {
            const unsigned cnt = hist[I];
            if (cnt == 0u) continue;
            const float* e = &S.tab[I][0];
            const unsigned ap = __float_as_uint(e[72]);
            float Y[NC][m];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const bool in = ((ap >> i) & 1u) != 0u;
#pragma unroll
                for (int c = 0; c < m; ++c) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NC; ++j) s = fmaf(e[8 * j + i], S.Jc[j][Model::act(c)], s);    // G[i][j] = M[i][j] on the set
                    Y[i][c] = in ? -s : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < n; ++k)
#pragma unroll
                for (int c = 0; c < m; ++c) {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < NC; ++i) s = fmaf(S.Jc[i][k], Y[i][c], s);
                    const float Bkc = (k == Model::act(c) ? 1.f : 0.f) - S.Dinv[k] * s;
                    acc[Model::perm(k) * m + c] += (float)cnt * Bkc;
                }
        }
        if ((poison_bits & 0x7f800000u) == 0x7f800000u) acc[0] = irs_poison();
        block_reduce_lds<P, NW>(acc, red);
    } else {
        block_reduce_lds<PI, NW>(acc, red);
        __syncthreads();
        for (int q = tid; q < PI; q += BLOCK) {
            float s = red[q];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w * PPI + q];
            toti[q] = s;
        }
        __syncthreads();
        // [Gram | du (f - xb)' | sum du]:  df_k = sum_c JD[k][c] lam_c - Db0_k + [k = act j] (D^-1 K)_j du_j
        float out = 0.f;
        if (tid < P) {
            if (tid < NG) out = toti[tid];
            else if (tid < NG + m * n) {
                const int i = (tid - NG) / n, kext = (tid - NG) % n;
                int k = 0, jact = -1;
#pragma unroll
                for (int kk = 0; kk < n; ++kk) k = (Model::perm(kk) == kext) ? kk : k;
#pragma unroll
                for (int j = 0; j < m; ++j) jact = (Model::act(j) == k) ? j : jact;
                float s = -S.Db0[k] * toti[NG + m * NC + i];
                for (int c = 0; c < NC; ++c) s = fmaf(S.JD[k][c], toti[NG + i * NC + c], s);
                if (jact >= 0) {
                    const int lo = i < jact ? i : jact, hi = i < jact ? jact : i;
                    s = fmaf(S.DK[jact], toti[lo * m - lo * (lo - 1) / 2 + (hi - lo)], s);
                }
                out = s;
            } else {
                out = toti[NG + m * NC + (tid - NG - m * n)];
            }
        }
        __syncthreads();
        for (int q = tid; q < NW * TR::PP; q += BLOCK) red[q] = 0.f;
        __syncthreads();
        if (tid < P) red[tid] = out;
    }
    UG_STAMP(4);
    smooth_finish<Model, MODE, FUSE, BLOCK, true>(a, red, red64, tot, fin, s_ticket, t, blk, tid, a.fnom + (size_t)t * n);
    UG_STAMP(5);
}

template <class Model, int MODE>
void ug_launch_m(const SmoothArgs& a, bool rng, bool fuse, hipStream_t st) {
    dim3 grid(a.nblk, a.T), block(kUgBlock);
    if (rng) {
        if (fuse) hipLaunchKernelGGL((smooth_ug_kernel<Model, MODE, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((smooth_ug_kernel<Model, MODE, true, false>), grid, block, 0, st, a);
    } else {
        if (fuse) hipLaunchKernelGGL((smooth_ug_kernel<Model, MODE, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((smooth_ug_kernel<Model, MODE, false, false>), grid, block, 0, st, a);
    }
}

}  // namespace

#ifdef IRS_UG_STAMPS
extern "C" int irs_debug_ug_stamps(unsigned long long* host_out, int n_wgs) {
    if (n_wgs > kUgStampWgs) n_wgs = kUgStampWgs;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ug_stamps), (size_t)n_wgs * kUgStampSlots * sizeof(unsigned long long));
}
#endif

bool irs_smooth_ug_supported(int model, int mode) {
    const char* e = getenv("IRS_UG");
    if (e != nullptr && atoi(e) == 0) return false;
    return model == IRS_MODEL_PLANAR_HAND_EXACT && (mode == IRS_SMOOTH_ZERO_ORDER_B || mode == IRS_SMOOTH_FIRST_ORDER);
}

// workgroups per timestep: one 512-thread workgroup per CU holds the 70 KB table, so the grid is about the CU count
// and every wave loops over its share of the 64-sample blocks; never more workgroups than blocks / 8
int irs_smooth_ug_nblk(int T, int N) {
    static int cus = 0;
    if (cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                  ? prop.multiProcessorCount : 256;
        if (cus < 1) cus = 256;
    }
    const int nblocks = (N + 63) / 64, NW = kUgBlock / 64;
    int nb = cus / (T < 1 ? 1 : T);
    if (nb < 1) nb = 1;
    const int by_work = (nblocks + 1 + NW - 1) / NW;     // + 1: the nominal wave
    if (nb > by_work) nb = by_work;
    return nb < 1 ? 1 : nb;
}

int irs_smooth_ug_launch(int model, int mode, const SmoothArgs& a, bool rng, bool fuse, hipStream_t st) {
    if (model != IRS_MODEL_PLANAR_HAND_EXACT) return IRS_ERR_UNSUPPORTED;
    if (mode == IRS_SMOOTH_ZERO_ORDER_B) ug_launch_m<PlanarHandExactModel, IRS_SMOOTH_ZERO_ORDER_B>(a, rng, fuse, st);
    else if (mode == IRS_SMOOTH_FIRST_ORDER) ug_launch_m<PlanarHandExactModel, IRS_SMOOTH_FIRST_ORDER>(a, rng, fuse, st);
    else return IRS_ERR_UNSUPPORTED;
    return IRS_OK;
}
