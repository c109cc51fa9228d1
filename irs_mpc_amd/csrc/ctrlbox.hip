// Quasistatic descent with ONE control box, solved exactly by an active-set method.
//
// IrsLqrQuasistatic.local_descent (irs_lqr/irs_lqr_quasistatic.py:286-345) re-solves, for every t,
// the tail QP of solve_tvlqr (irs_lqr/tv_lqr.py:30-137, position-controlled branch: input cost on
// du_t = u_t - u_{t-1}) and applies its first control to the true dynamics.  Every example of the
// reference bounds EITHER u_t (u_bounds_abs: a trust region around the nominal actuated
// positions) OR du_t (u_bounds_rel: a rate limit).  Writing the QP so that the bounded quantity is
// the control of an LQR with state s = [x; w], w = u_{t-1},
//     kind ABS: control u_t:         s+ = [[A,0],[0,0]] s + [B;I] u + [c;0],
//               stage (x-xd)'Q(x-xd) + (u-w)'R(u-w) = (s-sd)'diag(Q,R)(s-sd) + u'Ru + 2 s'Nc u, Nc=[0;-R]
//     kind REL: control v_t = du_t:  s+ = [[A,B],[0,I]] s + [B;I] v + [c;0],
//               stage (x-xd)'Q(x-xd) + v'Rv
// turns it into a control-box LQR: with the set of pinned control components fixed, the rest is an
// equality-constrained LQR solved by ONE backward sweep (policy u = K s + k, cost-to-go (P,p)),
// and that sweep does not depend on the start state.  So
//   * the active set and the sweep are carried from tail to tail: an unchanged active set costs a
//     forward vector sweep only, a changed one a PARTIAL backward sweep from the latest changed t;
//   * phase 1 updates the set primal-dual style (pin every violated bound, release every
//     wrong-signed multiplier at once: Hintermueller-Ito-Kunisch), which usually ends in a few
//     iterations but may cycle; after kPdasIter iterations phase 2 continues with the classic
//     primal active-set method (one constraint per iteration: blocking step or worst multiplier),
//     which is finite and monotone for a strictly convex QP.
// Restated in oracle/irs_oracle.py (quasistatic_ctrl_problem / ctrlbox_backward / ctrlbox_solve /
// local_descent_quasistatic_as), checked there against the ADMM solution of the same QPs.
//
// One solver wave (+ one plant wave), f64, everything in (dynamic) LDS: a latency-bound chain like the Riccati pass.
#include "boxqp.hpp"

namespace {

constexpr int kPdasIter = 10;
constexpr int KIND_ABS = 0, KIND_REL = 1;

__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// 1/d to ~1 ulp: hardware reciprocal + two Newton steps (a correctly rounded f64 divide is ~40
// dependent instructions, four times on the critical path of every backward step)
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmin(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = max(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = min(v, __shfl_xor(v, s, 64));
    return v;
}

// LDS record of one time step (doubles).  Record T holds only P, p.
template <int NR, int M>
struct CbLayout {
    static constexpr int NS = NR + M;
    static constexpr int oP = 0, op = oP + NS * NS, oK = op + NS, ok = oK + M * NS, oH = ok + M,
                         oG = oH + M * M, og = oG + M * NS, oA = og + M, oB = oA + NR * NR,
                         oc = oB + NR * M, ou = oc + NR, ous = ou + M, omu = ous + M,
                         oact = omu + M, olo = oact + M, ohi = olo + M, oqsd = ohi + M, S = oqsd + NR;
    static constexpr int NTRI = NS * (NS + 1) / 2;
    static constexpr int scratch = NS * NR + 2 * M * NS + 4 * NS + 2 * M * M + M * (NS + 1) + 2 * NR * NR +
                                   NTRI + M + 64;
    static __host__ __device__ size_t doubles(int T) { return (size_t)(T + 1) * S + scratch; }
};

// Two waves, two roles (as in ctrlbox_mfma.hip, which has the story): wave 0 solves, wave 1 -- the PLANT --
// owns the realised state, applies each tail's first control to the true (contact) dynamics, accumulates
// eval_cost and publishes the next start state; two workgroup barriers per tail.  The contact step used to
// be an out-of-line function called from the solver wave, and every value of the caller that was live
// across that call was at the mercy of the register pressure around it (256 VGPRs + up to 256 AGPRs taken
// by the callee): a wrong accumulated cost and an unwritten `info` were seen on the box-pivoting
// instantiation.  As a separate wave there is no call and nothing live across the step.
template <class Model, int KIND>
__global__ __launch_bounds__(128) void ctrlbox_descent_kernel(BoxArgs a) {
    constexpr int NR = Model::NX, M = Model::NU, NS = NR + M;
    constexpr double INF = __builtin_huge_val();
    static_assert(M * (NS + 1) <= 64 && M * NS + M <= 64, "one wave computes [K | k] in a single pass");
    using L = CbLayout<NR, M>;
    extern __shared__ double lds[];
    const int T = a.T, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // 0 solver, 1 plant
    double* F = lds;
    double* PB = F + (size_t)(T + 1) * L::S;           // NS x M    P B_
    double* WA = PB + NS * M;                          // NS x NR   P[:, :NR] A
    double* Y = WA + NS * NR;                          // M x NS
    double* qv = Y + M * NS;                           // NS
    double* hk = qv + NS;                              // M (padded NS)
    double* Hm = hk + NS;                              // M x M
    double* RHS = Hm + M * M;                          // M x (NS+1)
    double* Qsym = RHS + M * (NS + 1);                 // NR x NR
    double* Qdsym = Qsym + NR * NR;                    // NR x NR
    double* Rsym = Qdsym + NR * NR;                    // M x M
    double* sstart = Rsym + M * M;                     // NS   (plant -> solver)
    double* uctl = sstart + NS;                        // M: the tail's first control, clipped (solver -> plant)
    int* tri = reinterpret_cast<int*>(uctl + M);       // (i << 8 | j), i <= j, of the upper triangle

    auto rec_ = [&](int t) -> double* { return F + (size_t)t * L::S; };
    // problem data of the LQR in s = [x; w]
    auto Qs_ = [&](int i, int j) -> double {
        if (i < NR && j < NR) return Qsym[i * NR + j];
        if (KIND == KIND_ABS && i >= NR && j >= NR) return Rsym[(i - NR) * M + (j - NR)];
        return 0.0;
    };
    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // sentinels: a launch that does not reach its epilogue must not leave a previous launch's values
    // behind (info = -1 is rejected by the host like any other failure; cost = NaN)
    // ---- the plant wave: barriers S0 (tables up), then per tail A (start state ready), B (control ready) ----
    if (wave == 1) {
        double xr[NR], ur[M], xn[NR], up[M], ub[M];
        if (lane == 0 && a.cost) a.cost[0] = __builtin_nan("");
#pragma unroll
        for (int i = 0; i < NR; ++i) xr[i] = a.x0[i];
#pragma unroll
        for (int j = 0; j < M; ++j) up[j] = 0.0;
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) a.x_new[i] = xr[i];
        }
        auto quad = [&](const double* Wq, const double* e, int Kd) -> double {
            double q = 0.0;
            for (int i = 0; i < Kd; ++i)
                for (int j = 0; j < Kd; ++j) q += e[i] * Wq[i * Kd + j] * e[j];
            return q;
        };
        // start state [x; x[idx]]: each tail's first du is measured from the realised actuated
        // position (tv_lqr.py:99-100 at the tail's local t = 0)
        auto publish_start = [&]() {
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double v = xr[0];
#pragma unroll
                for (int i = 1; i < NR; ++i) v = (i == Model::u_into_x(j)) ? xr[i] : v;
                ub[j] = v;
            }
            if (lane < NS) {
                double v = xr[0];
#pragma unroll
                for (int i = 1; i < NR; ++i) v = (i == lane) ? xr[i] : v;
#pragma unroll
                for (int j = 0; j < M; ++j) v = (NR + j == lane) ? ub[j] : v;
                sstart[lane] = v;
            }
        };
        double cost = 0.0;
        unsigned warm = ~0u;                                // active set of the previous contact step
        irs_step_prepared<Model> pre;
        wg_barrier();                                       // S0
        publish_start();
        for (int tau = 0; tau < T; ++tau) {
            wg_barrier();                                   // A(tau)
            // while the solver wave works on this tail: everything of the coming contact step that depends on the
            // state alone (contact_models.hpp, irs_step_along_prepare)
            irs_step_along_prepare<Model>(a.p, xr, warm, pre);
            wg_barrier();                                   // B(tau)
#pragma unroll
            for (int j = 0; j < M; ++j) ur[j] = KIND == KIND_ABS ? uctl[j] : ub[j] + uctl[j];
            {   // IrsLqrQuasistatic.eval_cost (irs_lqr_quasistatic.py:153-194)
                double e[NR], dv[M];
#pragma unroll
                for (int i = 0; i < NR; ++i) e[i] = xr[i] - a.xd[(size_t)tau * NR + i];
#pragma unroll
                for (int j = 0; j < M; ++j) dv[j] = ur[j] - (tau == 0 ? ub[j] : up[j]);
                cost += quad(Qsym, e, NR) + quad(Rsym, dv, M);
            }
            irs_step_along_finish<Model>(a.p, xr, ur, pre, xn, &warm);
#pragma unroll
            for (int i = 0; i < NR; ++i) xr[i] = xn[i];
#pragma unroll
            for (int j = 0; j < M; ++j) up[j] = ur[j];
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < M; ++j) a.u_new[(size_t)tau * M + j] = ur[j];
#pragma unroll
                for (int i = 0; i < NR; ++i) a.x_new[(size_t)(tau + 1) * NR + i] = xr[i];
            }
            publish_start();
        }
        {
            double e[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) e[i] = xr[i] - a.xd[(size_t)T * NR + i];
            cost += quad(Qdsym, e, NR);
        }
        if (lane == 0 && a.cost) a.cost[0] = cost;
        return;
    }

    // ---- the solver wave: setup ----------------------------------------------------------
    if (lane == 0) {
        a.info[0] = -1; a.info[1] = -1; a.info[2] = -1;
    }
    for (int q = lane; q < NR * NR; q += 64) {
        int i = q / NR, j = q % NR;
        Qsym[q] = 0.5 * (a.Q[i * NR + j] + a.Q[j * NR + i]);
        Qdsym[q] = 0.5 * (a.Qd[i * NR + j] + a.Qd[j * NR + i]);
    }
    for (int q = lane; q < M * M; q += 64) {
        int i = q / M, j = q % M;
        Rsym[q] = 0.5 * (a.R[i * M + j] + a.R[j * M + i]);
    }
    for (int q = lane; q < L::NTRI; q += 64) {
        int i = 0, r = q;
        while (r >= NS - i) { r -= NS - i; ++i; }
        tri[q] = (i << 8) | (i + r);
    }
    wave_sync();
    const double* blo = KIND == KIND_ABS ? a.ulo : a.dlo;
    const double* bhi = KIND == KIND_ABS ? a.uhi : a.dhi;
    const int bs = KIND == KIND_ABS ? a.su : a.sd;
    for (int t = 0; t < T; ++t) {
        double* rec = rec_(t);
        for (int q = lane; q < NR * NR; q += 64) rec[L::oA + q] = a.At[(size_t)t * NR * NR + q];
        for (int q = lane; q < NR * M; q += 64) rec[L::oB + q] = a.Bt[(size_t)t * NR * M + q];
        if (lane < NR) {
            rec[L::oc + lane] = a.ct[(size_t)t * NR + lane];
            double sq = 0.0;
            for (int j = 0; j < NR; ++j) sq += Qsym[lane * NR + j] * a.xd[(size_t)t * NR + j];
            rec[L::oqsd + lane] = sq;                  // (Qs sd_t)[:NR]; the w block of sd is zero
        }
        if (lane < M) {
            const double lo_ = blo ? blo[(size_t)t * bs + lane] : -INF;
            const double hi_ = bhi ? bhi[(size_t)t * bs + lane] : INF;
            rec[L::olo + lane] = lo_;
            rec[L::ohi + lane] = hi_;
            // warm start of the first tail (the previous iLQR iteration's converged set), cleaned:
            // {-1, 0, +1}, and nothing pinned at an infinite bound
            double a0 = a.act_io ? a.act_io[(size_t)t * M + lane] : 0.0;
            a0 = a0 < 0.0 ? (lo_ > -INF ? -1.0 : 0.0) : (a0 > 0.0 ? (hi_ < INF ? 1.0 : 0.0) : 0.0);
            rec[L::oact + lane] = a0;
            rec[L::ou + lane] = 0.0;
            rec[L::ous + lane] = 0.0;
            rec[L::omu + lane] = 0.0;
        }
    }
    wave_sync();
    {   // terminal cost-to-go: P_T = Qsd, p_T = -Qsd sd_T  (Qsd = diag(Qd, 0))
        double* rec = rec_(T);
        for (int q = lane; q < NS * NS; q += 64) {
            int i = q / NS, j = q % NS;
            rec[L::oP + q] = (i < NR && j < NR) ? Qdsym[i * NR + j] : 0.0;
        }
        if (lane < NS) {
            double s = 0.0;
            if (lane < NR)
                for (int j = 0; j < NR; ++j) s -= Qdsym[lane * NR + j] * a.xd[(size_t)T * NR + j];
            rec[L::op + lane] = s;
        }
    }
    wave_sync();

    int bad = 0;
    // ---- one backward step: policy (K,k) for the pinned set of step t, cost-to-go (P,p)_t ----
    auto backward_step = [&](int t) {
        double* rec = rec_(t);
        const double* nxt = rec_(t + 1);
        const double* P = nxt + L::oP;
        const double* p = nxt + L::op;
        const double* A = rec + L::oA;                 // NR x NR
        const double* B = rec + L::oB;                 // NR x M
        // The block structure of A_ = [[A, B or 0], [0, I or 0]] and B_ = [B; I] is used throughout:
        // every product below is a length-NR contraction plus at most one extra term.
        // ---- phase 1: WA = P[:, :NR] A ; PB = P B_ ; qv = P c_ + p
        for (int q = lane; q < NS * NR; q += 64) {
            const int i = q / NR, j = q % NR;
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < NR; ++l) s += P[i * NS + l] * A[l * NR + j];
            WA[q] = s;
        }
        for (int q = lane; q < NS * M; q += 64) {
            const int i = q / M, j = q % M;
            double s = P[i * NS + NR + j];
#pragma unroll
            for (int l = 0; l < NR; ++l) s += P[i * NS + l] * B[l * M + j];
            PB[q] = s;
        }
        if (lane < NS) {
            double s = p[lane];
#pragma unroll
            for (int l = 0; l < NR; ++l) s += P[lane * NS + l] * rec[L::oc + l];
            qv[lane] = s;
        }
        wave_sync();
        // ---- phase 2, one pass, one code shape: H = Ru + B_'PB (M*M items), g = B_'qv (M items),
        //      G[:, :NR] = PB[:NR]'A (M*NR items):  dst = extra + sum_{l<NR} X[l sx] Y[l sy]
        {
            static_assert(M * M + M + M * NR <= 64, "phase 2 is a single pass");
            const double *X = B, *Yp = PB;
            int sx = 0, sy = 0;
            double extra = 0.0;
            double* dst = nullptr;
            if (lane < M * M) {
                const int i = lane / M, j = lane % M;
                X = B + i; sx = M; Yp = PB + j; sy = M;
                extra = Rsym[lane] + PB[(NR + i) * M + j];
                dst = rec + L::oH + lane;
            } else if (lane < M * M + M) {
                const int i = lane - M * M;
                X = B + i; sx = M; Yp = qv; sy = 1;
                extra = qv[NR + i];
                dst = rec + L::og + i;
            } else if (lane < M * M + M + M * NR) {
                const int q = lane - (M * M + M), i = q / NR, j = q % NR;
                X = PB + i; sx = M; Yp = A + j; sy = NR;
                dst = rec + L::oG + i * NS + j;
            }
            double s = extra;
#pragma unroll
            for (int l = 0; l < NR; ++l) s += X[l * sx] * Yp[l * sy];
            if (dst) *dst = s;
        }
        wave_sync();
        // ---- phase 3: G[:, NR:] (= H' - Ru for REL, -Ru for ABS), masked system
        auto Ghi = [&](int i, int j) -> double {      // G[i][NR + j]
            return KIND == KIND_REL ? rec[L::oH + j * M + i] - Rsym[j * M + i] : -Rsym[i * M + j];
        };
        for (int q = lane; q < M * M; q += 64) {
            const int i = q / M, j = q % M;
            rec[L::oG + i * NS + NR + j] = Ghi(i, j);
            const bool fi = rec[L::oact + i] == 0.0, fj = rec[L::oact + j] == 0.0;
            Hm[q] = (fi && fj) ? rec[L::oH + q] : (i == j ? 1.0 : 0.0);
        }
        for (int q = lane; q < M * (NS + 1); q += 64) {
            const int i = q / (NS + 1), j = q % (NS + 1);
            const double ai = rec[L::oact + i];
            double s;
            if (ai == 0.0) {
                if (j < NR) s = -rec[L::oG + i * NS + j];
                else if (j < NS) s = -Ghi(i, j - NR);
                else {
                    s = -rec[L::og + i];
#pragma unroll
                    for (int l = 0; l < M; ++l) {
                        const double al = rec[L::oact + l];
                        const double ubl = al < 0.0 ? rec[L::olo + l] : rec[L::ohi + l];
                        s -= al != 0.0 ? rec[L::oH + i * M + l] * ubl : 0.0;
                    }
                }
            } else {
                s = j < NS ? 0.0 : (ai < 0.0 ? rec[L::olo + i] : rec[L::ohi + i]);
            }
            RHS[q] = s;
        }
        wave_sync();
        // ---- phase 4: Hm^-1 by LDL' in registers (every lane); lane q computes entry q of [K | k]
        {
            double Lm[M][M], Dg[M], Dinv[M];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double dj = Hm[j * M + j];
#pragma unroll
                for (int l = 0; l < j; ++l) dj -= Lm[j][l] * Lm[j][l] * Dg[l];
                if (!(dj > 0.0) && bad == 0) bad = t + 1;
                Dg[j] = dj;
                Dinv[j] = fast_rcp(dj);
#pragma unroll
                for (int i = j + 1; i < M; ++i) {
                    double s = Hm[i * M + j];
#pragma unroll
                    for (int l = 0; l < j; ++l) s -= Lm[i][l] * Lm[j][l] * Dg[l];
                    Lm[i][j] = s * Dinv[j];
                }
            }
            const int q = lane < M * (NS + 1) ? lane : 0;
            const int col = q / (NS + 1), j = q % (NS + 1);
            double y[M];                               // column `col` of Hm^-1 (= its row: symmetric)
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double s = (i == col) ? 1.0 : 0.0;
#pragma unroll
                for (int l = 0; l < i; ++l) s -= Lm[i][l] * y[l];
                y[i] = s;
            }
#pragma unroll
            for (int i = M - 1; i >= 0; --i) {
                double s = y[i] * Dinv[i];
#pragma unroll
                for (int l = i + 1; l < M; ++l) s -= Lm[l][i] * y[l];
                y[i] = s;
            }
            if (lane < M * (NS + 1)) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < M; ++l) s += y[l] * RHS[l * (NS + 1) + j];
                if (j < NS) rec[L::oK + col * NS + j] = s;
                else rec[L::ok + col] = s;
            }
        }
        wave_sync();
        // ---- phase 5: Y = H K + G ; hk = H k + g
        if (lane < M * NS) {
            const int i = lane / NS;
            double s = rec[L::oG + lane];
#pragma unroll
            for (int l = 0; l < M; ++l) s += rec[L::oH + i * M + l] * rec[L::oK + l * NS + lane % NS];
            Y[lane] = s;
        } else if (lane < M * NS + M) {
            const int i = lane - M * NS;
            double s = rec[L::og + i];
#pragma unroll
            for (int l = 0; l < M; ++l) s += rec[L::oH + i * M + l] * rec[L::ok + l];
            hk[i] = s;
        }
        wave_sync();
        // ---- phase 6: upper triangle of P_t = Qs + A_'W + K'Y + G'K (mirrored on store) and
        //      p_t = -Qs sd_t + A_'qv + K'hk + G'k, with W = P A_ = [WA | PB or 0]
        for (int it = lane; it < L::NTRI + NS; it += 64) {
            const bool isP = it < L::NTRI;
            const int i = isP ? (tri[it] >> 8) : it - L::NTRI;
            const int j = isP ? (tri[it] & 255) : 0;
            // column i of A_ restricted to its first NR rows, and whether A_[i][i] = 1 (REL, i >= NR)
            const double* Xc = i < NR ? A + i : B + (i - NR);
            const int sx = i < NR ? NR : M;
            const double xz = (i < NR || KIND == KIND_REL) ? 1.0 : 0.0;
            const bool diag = KIND == KIND_REL && i >= NR;
            // column j of W (or qv for the p row)
            const double* Wc = !isP ? qv : (j < NR ? WA + j : PB + (j - NR));
            const int sw = !isP ? 1 : (j < NR ? NR : M);
            const double wz = (!isP || j < NR || KIND == KIND_REL) ? 1.0 : 0.0;
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < NR; ++l) s += Xc[l * sx] * Wc[l * sw];
            if (diag) s += Wc[i * sw];
            s *= xz * wz;
            if (isP) {
#pragma unroll
                for (int l = 0; l < M; ++l)
                    s += rec[L::oK + l * NS + i] * Y[l * NS + j] + rec[L::oG + l * NS + i] * rec[L::oK + l * NS + j];
                s += Qs_(i, j);
                rec[L::oP + i * NS + j] = s;
                rec[L::oP + j * NS + i] = s;
            } else {
#pragma unroll
                for (int l = 0; l < M; ++l)
                    s += rec[L::oK + l * NS + i] * hk[l] + rec[L::oG + l * NS + i] * rec[L::ok + l];
                if (i < NR) s -= rec[L::oqsd + i];
                rec[L::op + i] = s;
            }
        }
        wave_sync();
    };

    // ---- policy rollout on the linear model from sstart: controls -> record offset `dst`, mu ----
    // The state never leaves registers: every lane carries all of s; per step lane j < M forms
    // control j, readlane broadcasts it, lanes M..M+NR-1 form the next x rows, readlane broadcasts
    // them.  The coefficient rows of step t+1 are fetched from LDS while step t computes, so the
    // dependent chain per step is two length-NS FMA chains and 2 (M + NR) readlanes -- no LDS round
    // trip.  Outputs (controls -> record offset `dst`, multipliers -> omu) are stored off the chain.
    auto bcast = [&](double v, int src) -> double {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    };
    struct Rows { double r1[NS], r2[NS], r3[M], c1, c2; };
    auto load_rows = [&](int t, Rows& R) {
        const double* rec = rec_(t);
        const bool isu = lane < M;                       // a control lane, else an x-row lane (clamped)
        const int i = isu ? lane : (lane - M < NR ? lane - M : 0);
#pragma unroll
        for (int l = 0; l < NS; ++l) {
            R.r1[l] = isu ? rec[L::oK + i * NS + l] : (l < NR ? rec[L::oA + i * NR + l] : 0.0);
            R.r2[l] = isu ? rec[L::oG + i * NS + l] : 0.0;
        }
#pragma unroll
        for (int l = 0; l < M; ++l) R.r3[l] = isu ? rec[L::oH + i * M + l] : rec[L::oB + i * M + l];
        R.c1 = isu ? rec[L::ok + i] : rec[L::oc + i];
        R.c2 = isu ? rec[L::og + i] : 0.0;
    };
    auto policy_rollout = [&](int t0, int dst) {
        double sr[NS];
#pragma unroll
        for (int l = 0; l < NS; ++l) sr[l] = sstart[l];
        Rows cur, nxt;
        load_rows(t0, cur);
        for (int t = t0; t < T; ++t) {
            if (t + 1 < T) load_rows(t + 1, nxt);
            // controls (lanes < M), then broadcast
            double acc = cur.c1;
#pragma unroll
            for (int l = 0; l < NS; ++l) acc += cur.r1[l] * sr[l];      // control lane: K_j s + k_j; x lane: A_i x + c_i
            double v[M];
#pragma unroll
            for (int j = 0; j < M; ++j) v[j] = bcast(acc, j);
            double ua[M];                                                // absolute command
#pragma unroll
            for (int j = 0; j < M; ++j) ua[j] = v[j] + (KIND == KIND_REL ? sr[NR + j] : 0.0);
            // multipliers (control lanes, off the chain) and next x rows (x lanes)
            double tail = 0.0;
#pragma unroll
            for (int l = 0; l < M; ++l) tail += cur.r3[l] * (lane < M ? v[l] : ua[l]);   // H_j v  |  B_i u_abs
            if (lane < M) {
                double mu = cur.c2 + tail;
#pragma unroll
                for (int l = 0; l < NS; ++l) mu += cur.r2[l] * sr[l];
                double* rec = rec_(t);
                rec[dst + lane] = acc;
                rec[L::omu + lane] = mu;
            }
            const double xrow = acc + tail;                              // meaningful on lanes M..M+NR-1
#pragma unroll
            for (int i = 0; i < NR; ++i) sr[i] = bcast(xrow, M + i);
#pragma unroll
            for (int j = 0; j < M; ++j) sr[NR + j] = ua[j];
            cur = nxt;
        }
        wave_sync();
    };

    // ---- MPC loop (solver side) ----------------------------------------------------------
    int it_max = 0, n_fail = 0;
    const double tol = a.eps;
    bool full = true;                                  // no valid backward sweep yet
    wg_barrier();                                      // S0: tables and records are up

    for (int tau = 0; tau < T; ++tau) {
        wg_barrier();                                  // A(tau): the plant has published this tail's start state
        const int t0 = tau;
        int t_dirty = full ? T - 1 : t0 - 1;           // the sweep of the previous tail covers t >= tau
        int iters = 0;
        bool conv = false;
        // ---- phase 1: primal-dual active set
        for (int it = 0; it < kPdasIter && !conv; ++it) {
            ++iters;
            for (int t = t_dirty; t >= t0; --t) backward_step(t);
            policy_rollout(t0, L::ou);
            int chg = -1;
            for (int q = t0 * M + lane; q < T * M; q += 64) {
                const int t = q / M, j = q % M;
                double* rec = rec_(t);
                const double ac = rec[L::oact + j], u = rec[L::ou + j], mu = rec[L::omu + j];
                double nw = ac;
                if (ac == 0.0) {
                    if (u < rec[L::olo + j] - tol) nw = -1.0;
                    else if (u > rec[L::ohi + j] + tol) nw = 1.0;
                } else if (ac < 0.0) {
                    if (mu < -tol) nw = 0.0;
                } else {
                    if (mu > tol) nw = 0.0;
                }
                if (nw != ac) { rec[L::oact + j] = nw; chg = max(chg, t); }
            }
            chg = wave_max_i(chg);
            wave_sync();
            if (chg < 0) conv = true;
            else t_dirty = chg;
        }
        // ---- phase 2: primal active set from the clipped iterate
        if (!conv) {
            int chg = -1;
            for (int q = t0 * M + lane; q < T * M; q += 64) {
                const int t = q / M, j = q % M;
                double* rec = rec_(t);
                const double lo = rec[L::olo + j], hi = rec[L::ohi + j];
                const double u = fmin(fmax(rec[L::ou + j], lo), hi);
                rec[L::ou + j] = u;
                const double nw = u <= lo ? -1.0 : (u >= hi ? 1.0 : 0.0);
                if (nw != rec[L::oact + j]) { rec[L::oact + j] = nw; chg = max(chg, t); }
            }
            chg = wave_max_i(chg);
            wave_sync();
            t_dirty = max(t_dirty, chg);
            for (int it2 = 0; it2 < a.max_iter && !conv; ++it2) {
                ++iters;
                for (int t = t_dirty; t >= t0; --t) backward_step(t);
                t_dirty = t0 - 1;
                policy_rollout(t0, L::ous);
                // largest feasible step along d = us - u over the free components
                double best = INF;
                int bq = 0x7fffffff;
                for (int q = t0 * M + lane; q < T * M; q += 64) {
                    const int t = q / M, j = q % M;
                    const double* rec = rec_(t);
                    if (rec[L::oact + j] == 0.0) {
                        const double u = rec[L::ou + j], d = rec[L::ous + j] - u;
                        double room = INF;
                        if (d > 0.0) room = (rec[L::ohi + j] - u) / d;
                        else if (d < 0.0) room = (rec[L::olo + j] - u) / d;
                        if (room < best) { best = room; bq = q; }
                    }
                }
                const double alpha = wave_min(best);
                if (alpha < 1.0) {
                    const int qb = wave_min_i(best == alpha ? bq : 0x7fffffff);
                    for (int q = t0 * M + lane; q < T * M; q += 64) {
                        const int t = q / M, j = q % M;
                        double* rec = rec_(t);
                        const double u = rec[L::ou + j], d = rec[L::ous + j] - u;
                        if (q == qb) {
                            rec[L::oact + j] = d > 0.0 ? 1.0 : -1.0;
                            rec[L::ou + j] = d > 0.0 ? rec[L::ohi + j] : rec[L::olo + j];
                        } else {
                            rec[L::ou + j] = u + alpha * d;
                        }
                    }
                    wave_sync();
                    t_dirty = qb / M;
                    continue;
                }
                // full step: u = us; optimal if every pinned multiplier has the right sign
                double worst = 0.0;
                int wq = 0x7fffffff;
                for (int q = t0 * M + lane; q < T * M; q += 64) {
                    const int t = q / M, j = q % M;
                    double* rec = rec_(t);
                    rec[L::ou + j] = rec[L::ous + j];
                    const double ac = rec[L::oact + j], mu = rec[L::omu + j];
                    const double viol = ac < 0.0 ? -mu : (ac > 0.0 ? mu : 0.0);
                    if (viol > worst) { worst = viol; wq = q; }
                }
                const double wmax = wave_max(worst);
                if (wmax <= tol) {
                    wave_sync();
                    conv = true;
                } else {
                    const int qw = wave_min_i(worst == wmax ? wq : 0x7fffffff);
                    if (lane == 0) rec_(qw / M)[L::oact + qw % M] = 0.0;
                    wave_sync();
                    t_dirty = qw / M;
                }
            }
        }
        it_max = max(it_max, iters);
        n_fail += conv ? 0 : 1;
        full = !conv;
        if (tau == 0 && a.act_io != nullptr) {
            for (int q = lane; q < T * M; q += 64) a.act_io[q] = rec_(q / M)[L::oact + q % M];
        }
        // first control of the tail solution (clipped) -> the plant
        if (lane < M) {
            const double* rec = rec_(tau);
            uctl[lane] = fmin(fmax(rec[L::ou + lane], rec[L::olo + lane]), rec[L::ohi + lane]);
        }
        wg_barrier();                                  // B(tau)
    }
    if (lane == 0) {
        a.info[0] = bad; a.info[1] = it_max; a.info[2] = n_fail;
    }
}

template <class Model, int KIND>
int launch_ctrlbox(const BoxArgs& a, hipStream_t st) {
    const size_t bytes = CbLayout<Model::NX, Model::NU>::doubles(a.T) * sizeof(double);
    if (bytes > 160 * 1024 - 512) {
        irs_set_error("irs_quasistatic_box_descent: horizon T=%d needs %zu bytes of LDS (max ~160 KB)", a.T, bytes);
        return IRS_ERR_UNSUPPORTED;
    }
    auto kern = ctrlbox_descent_kernel<Model, KIND>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        irs_set_error("irs_quasistatic_box_descent: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return IRS_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(128), bytes, st, a);
    return IRS_OK;
}

}  // namespace

int irs_ctrlbox_launch(int model, const BoxArgs& a, int kind, hipStream_t st) {
    int rc = IRS_ERR_UNSUPPORTED;
    IRS_DISPATCH_MODEL(model, {
        if constexpr (has_u_into_x<Model>::value) {
            rc = kind == KIND_ABS ? launch_ctrlbox<Model, KIND_ABS>(a, st) : launch_ctrlbox<Model, KIND_REL>(a, st);
        } else {
            irs_set_error("irs_quasistatic_box_descent: model %d is not position controlled", model);
        }
    });
    return rc;
}

size_t irs_ctrlbox_lds_bytes(int model, int T) {
    size_t r = 0;
    IRS_DISPATCH_MODEL(model, {
        if constexpr (has_u_into_x<Model>::value) r = CbLayout<Model::NX, Model::NU>::doubles(T) * sizeof(double);
    });
    return r;
}
