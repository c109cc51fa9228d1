// Box-constrained TV-LQR: solve_tvlqr with ACTIVE abs bounds (irs_lqr/tv_lqr.py:112-123)
// inside the MPC loop of IrsLqr.local_descent (irs_lqr/irs_lqr.py:169-184): for every t the
// tail QP over t..T is re-solved from the realised state and only its first control is
// applied to the TRUE dynamics.
//
// The reference hands each QP to OSQP.  Here: ADMM on the box, with the equality-
// constrained (LQR) sub-problem solved exactly by a Riccati sweep,
//     z <- argmin f(z) + rho/2 |z - w + y|^2 ,  w <- clip(a z + (1-a) w + y) ,  y <- y + ... - w,
// where only bounded components carry a rho term.  The Riccati matrices depend on
// (A,B,Q,R,rho) but not on the linear terms, so ONE backward factorisation (kept in LDS)
// serves every ADMM iteration of every one of the T tail re-solves; an ADMM iteration is
// then two vector sweeps over the horizon.  Successive re-solves are warm started.
// Restated in oracle/irs_oracle.py (tvlqr_box_factor / tvlqr_box_solve / local_descent_box),
// whose solutions are certified against the QP's KKT conditions.
//
//
// Position-controlled (quasistatic) variant, DU = true: IrsLqrQuasistatic.local_descent
// (irs_lqr/irs_lqr_quasistatic.py:286-345) calls solve_tvlqr with indices_u_into_x, whose cost is
// on du_t = u_t - u_{t-1} (du_0 = u_0 - x_0[idx], irs_lqr/tv_lqr.py:98-108, full R: alpha = 1) and
// whose bounds are per-time trust regions (x_bound_abs, u_bound_abs) and rate limits
// (u_bound_rel).  That QP is the SAME box-LQR in the augmented state z = [x; u_prev] with control
// v = du:  z+ = [[A,B],[0,I]] z + [B;I] v + [c;0],  cost (x-xd)'Q(x-xd) + v'Rv,  box on z
// (x bounds; u bounds = bounds on the u_prev block one step later) and on v.  The augmented
// matrices are never materialised: accessors below read (A,B,c) directly.
//
// One wave, f64, everything in (dynamic) LDS: a latency-bound chain like the Riccati pass.
#include "boxqp.hpp"

namespace {

__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s, 64));
    return v;
}

template <int N, int M>
struct BoxLayout {
    // per-timestep factor record
    static constexpr int oAcl = 0, oK = oAcl + N * N, oMinv = oK + M * N, oHinv = oMinv + M * N,
                         oB = oHinv + M * M, oC = oB + N * M, oD = oC + N, oQx = oD + N, S = oQx + N;
    static __host__ __device__ size_t doubles(int T) {
        // factor records + qx_T + wx,yx,zx (T+1,N) + wu,yu,zu,k (T,M) + scratch
        return (size_t)T * S + N + 3 * (size_t)(T + 1) * N + 4 * (size_t)T * M + 4 * N * N + 8 * N + 4 * M + 64;
    }
};

template <class Model, bool DU>
__global__ __launch_bounds__(64) void box_descent_kernel(BoxArgs a) {
    if (a.run_flag != nullptr && *a.run_flag == 0) return;          // uniform
    constexpr int NR = Model::NX, M = Model::NU;      // real state / control sizes
    constexpr int N = NR + (DU ? M : 0);              // size of the QP's state (z = [x; u_prev] if DU)
    constexpr double INF = __builtin_huge_val();
    using L = BoxLayout<N, M>;
    // augmented problem data, read straight from the caller's (A, B, c, Q, Qd, xd)
    auto A_ = [&](int t, int i, int j) -> double {
        if (i < NR) return j < NR ? a.At[((size_t)t * NR + i) * NR + j] : a.Bt[((size_t)t * NR + i) * M + (j - NR)];
        return i == j ? 1.0 : 0.0;
    };
    auto B_ = [&](int t, int i, int j) -> double {
        if (i < NR) return a.Bt[((size_t)t * NR + i) * M + j];
        return (i - NR) == j ? 1.0 : 0.0;
    };
    auto c_ = [&](int t, int i) -> double { return i < NR ? a.ct[(size_t)t * NR + i] : 0.0; };
    auto xd_ = [&](int t, int i) -> double { return i < NR ? a.xd[(size_t)t * NR + i] : 0.0; };
    // bounds of the QP's state component i at time t, and of its control component j
    auto zlo_ = [&](int t, int i) -> double {
        if (i < NR) return a.xlo ? a.xlo[(size_t)t * a.sx + i] : -INF;
        return (a.ulo && t >= 1) ? a.ulo[(size_t)(t - 1) * a.su + (i - NR)] : -INF;
    };
    auto zhi_ = [&](int t, int i) -> double {
        if (i < NR) return a.xhi ? a.xhi[(size_t)t * a.sx + i] : INF;
        return (a.uhi && t >= 1) ? a.uhi[(size_t)(t - 1) * a.su + (i - NR)] : INF;
    };
    auto vlo_ = [&](int t, int j) -> double {
        if (DU) return a.dlo ? a.dlo[(size_t)t * a.sd + j] : -INF;
        return a.ulo ? a.ulo[(size_t)t * a.su + j] : -INF;
    };
    auto vhi_ = [&](int t, int j) -> double {
        if (DU) return a.dhi ? a.dhi[(size_t)t * a.sd + j] : INF;
        return a.uhi ? a.uhi[(size_t)t * a.su + j] : INF;
    };
    extern __shared__ double lds[];
    const int T = a.T, lane = threadIdx.x;
    double* F = lds;                                   // T records of L::S doubles
    double* qxT = F + (size_t)T * L::S;                // Qd xd_T
    double* wx = qxT + N;                              // (T+1, N)
    double* yx = wx + (size_t)(T + 1) * N;
    double* zx = yx + (size_t)(T + 1) * N;
    double* wu = zx + (size_t)(T + 1) * N;             // (T, M)
    double* yu = wu + (size_t)T * M;
    double* zu = yu + (size_t)T * M;
    double* kk = zu + (size_t)T * M;
    double* P = kk + (size_t)T * M;                    // scratch: P, A, W (N x N), vectors
    double* Am = P + N * N;
    double* Wm = Am + N * N;
    double* PB = Wm + N * N;                           // N x M  (fits in N*N)
    double* pv = PB + N * N;                           // p (N)
    double* gv = pv + N;                               // g (N)
    double* sv = gv + N;                               // s (M) .. padded to N
    double* mxv = sv + N;                              // bounded masks (any finite bound at any t)
    double* muv = mxv + N;                             // (M)
    double* Hs = muv + M;                              // M x M scratch (<= 16)

    // ---- setup ------------------------------------------------------------------------
    if (lane < N) {
        bool any = false;
        for (int t = 1; t <= T; ++t) any = any || isfinite(zlo_(t, lane)) || isfinite(zhi_(t, lane));
        mxv[lane] = any ? 1.0 : 0.0;
    }
    if (lane < M) {
        bool any = false;
        for (int t = 0; t < T; ++t) any = any || isfinite(vlo_(t, lane)) || isfinite(vhi_(t, lane));
        muv[lane] = any ? 1.0 : 0.0;
    }
    for (int q = lane; q < (T + 1) * N; q += 64) { wx[q] = 0.0; yx[q] = 0.0; zx[q] = 0.0; }
    for (int q = lane; q < T * M; q += 64) { wu[q] = 0.0; yu[q] = 0.0; zu[q] = 0.0; kk[q] = 0.0; }
    wave_sync();
    const double hr = 0.5 * a.rho;
    auto qs = [&](const double* Qm, int i, int j) -> double {
        return (i < NR && j < NR) ? 0.5 * (Qm[i * NR + j] + Qm[j * NR + i]) : 0.0;
    };
    // P_T = Qd + rho/2 Mx ; qx_T = Qd xd_T
    for (int q = lane; q < N * N; q += 64) {
        int i = q / N, j = q % N;
        P[q] = qs(a.Qd, i, j) + (i == j ? hr * mxv[i] : 0.0);
    }
    if (lane < N) {
        double s = 0.0;
        for (int j = 0; j < N; ++j) s += qs(a.Qd, lane, j) * xd_(T, j);
        qxT[lane] = s;
    }
    wave_sync();

    // ---- factorisation: backward Riccati with Q^ = Q + rho/2 Mx, R^ = alpha R + rho/2 Mu --
    int bad = 0;
    for (int t = T - 1; t >= 0; --t) {
        double* rec = F + (size_t)t * L::S;
        for (int q = lane; q < N * N; q += 64) Am[q] = A_(t, q / N, q % N);
        for (int q = lane; q < N * M; q += 64) rec[L::oB + q] = B_(t, q / M, q % M);
        if (lane < N) {
            rec[L::oC + lane] = c_(t, lane);
            double s = 0.0;
            for (int j = 0; j < N; ++j) s += qs(a.Q, lane, j) * xd_(t, j);
            rec[L::oQx + lane] = s;
        }
        wave_sync();
        const double* B = rec + L::oB;
        // PB = P B ; d = P c
        for (int q = lane; q < N * M; q += 64) {
            int i = q / M, j = q % M;
            double s = 0.0;
            for (int l = 0; l < N; ++l) s += P[i * N + l] * B[l * M + j];
            PB[q] = s;
        }
        if (lane < N) {
            double s = 0.0;
            for (int l = 0; l < N; ++l) s += P[lane * N + l] * rec[L::oC + l];
            rec[L::oD + lane] = s;
        }
        wave_sync();
        // H = R^ + B'PB
        for (int q = lane; q < M * M; q += 64) {
            int i = q / M, j = q % M;
            double s = 0.5 * a.alpha * (a.R[i * M + j] + a.R[j * M + i]) + (i == j ? hr * muv[i] : 0.0);
            for (int l = 0; l < N; ++l) s += B[l * M + i] * PB[l * M + j];
            Hs[q] = s;
        }
        wave_sync();
        // H^-1 by LDL' in registers (every lane), lane j < M keeps column j
        {
            double Lm[M][M], Dg[M], Dinv[M];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double dj = Hs[j * M + j];
#pragma unroll
                for (int l = 0; l < j; ++l) dj -= Lm[j][l] * Lm[j][l] * Dg[l];
                if (!(dj > 0.0) && bad == 0) bad = t + 1;
                Dg[j] = dj;
                Dinv[j] = 1.0 / dj;
#pragma unroll
                for (int i = j + 1; i < M; ++i) {
                    double s = Hs[i * M + j];
#pragma unroll
                    for (int l = 0; l < j; ++l) s -= Lm[i][l] * Lm[j][l] * Dg[l];
                    Lm[i][j] = s * Dinv[j];
                }
            }
            if (lane < M) {
                double y[M];
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    double s = (i == lane) ? 1.0 : 0.0;
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
#pragma unroll
                for (int i = 0; i < M; ++i) rec[L::oHinv + i * M + lane] = y[i];
            }
        }
        wave_sync();
        // Minv = H^-1 B' (M x N)
        for (int q = lane; q < M * N; q += 64) {
            int i = q / N, j = q % N;
            double s = 0.0;
            for (int l = 0; l < M; ++l) s += rec[L::oHinv + i * M + l] * B[j * M + l];
            rec[L::oMinv + q] = s;
        }
        // W = P A
        for (int q = lane; q < N * N; q += 64) {
            int i = q / N, j = q % N;
            double s = 0.0;
            for (int l = 0; l < N; ++l) s += P[i * N + l] * Am[l * N + j];
            Wm[q] = s;
        }
        wave_sync();
        // K = -Minv W  (= -H^-1 B'P A)
        for (int q = lane; q < M * N; q += 64) {
            int i = q / N, j = q % N;
            double s = 0.0;
            for (int l = 0; l < N; ++l) s -= rec[L::oMinv + i * N + l] * Wm[l * N + j];
            rec[L::oK + q] = s;
        }
        wave_sync();
        // Acl = A + B K
        for (int q = lane; q < N * N; q += 64) {
            int i = q / N, j = q % N;
            double s = Am[q];
            for (int l = 0; l < M; ++l) s += B[i * M + l] * rec[L::oK + l * N + j];
            rec[L::oAcl + q] = s;
        }
        wave_sync();
        // P <- Q^ + sym(W' Acl)   (W' Acl = A'P Acl)
        double pn[(N * N + 63) / 64];
#pragma unroll
        for (int r = 0; r < (N * N + 63) / 64; ++r) {
            int q = lane + 64 * r;
            pn[r] = 0.0;
            if (q < N * N) {
                int i = q / N, j = q % N;
                double s = 0.0, s2 = 0.0;
                for (int l = 0; l < N; ++l) {
                    s += Wm[l * N + i] * rec[L::oAcl + l * N + j];
                    s2 += Wm[l * N + j] * rec[L::oAcl + l * N + i];
                }
                pn[r] = qs(a.Q, i, j) + (i == j ? hr * mxv[i] : 0.0) + 0.5 * (s + s2);
            }
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < (N * N + 63) / 64; ++r) {
            int q = lane + 64 * r;
            if (q < N * N) P[q] = pn[r];
        }
        wave_sync();
    }

    // ---- MPC loop: T tail re-solves, first control applied to the true dynamics ----------
    double xr[NR], ur[M], xn[NR], up[M];
#pragma unroll
    for (int i = 0; i < NR; ++i) xr[i] = a.x0[i];
#pragma unroll
    for (int j = 0; j < M; ++j) up[j] = 0.0;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) a.x_new[i] = xr[i];
    }
    double cost = 0.0;
    auto quad = [&](const double* Wm_, const double* e, int K) -> double {   // e' sym(W) e, K = NR or M
        double s = 0.0;
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) s += e[i] * Wm_[i * K + j] * e[j];
        return s;
    };
    int it_max = 0, n_fail = 0;
    const double al = a.relax;
    for (int tau = 0; tau < T; ++tau) {
        // the tail problem starts from the realised state; for DU its u_prev block is the
        // realised actuated position x_tau[idx] (tv_lqr.py:99-100 at the tail's local t = 0)
        double ub[M];
#pragma unroll
        for (int j = 0; j < M; ++j) ub[j] = 0.0;
        if constexpr (DU) {
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double v = xr[0];
#pragma unroll
                for (int i = 1; i < NR; ++i) v = (i == Model::u_into_x(j)) ? xr[i] : v;
                ub[j] = v;
            }
        }
        if (lane < N) {
            double v = xr[0];
#pragma unroll
            for (int i = 1; i < NR; ++i) v = (i == lane) ? xr[i] : v;
#pragma unroll
            for (int j = 0; j < M; ++j) v = (NR + j == lane) ? ub[j] : v;
            zx[(size_t)tau * N + lane] = v;
        }
        wave_sync();
        int it = 0;
        bool conv = false;
        while (it < a.max_iter && !conv) {
            ++it;
            // backward affine sweep: p_T = -(Qd xd_T + rho/2 mx (w - y)_T)
            if (lane < N) pv[lane] = -(qxT[lane] + hr * mxv[lane] * (wx[(size_t)T * N + lane] - yx[(size_t)T * N + lane]));
            wave_sync();
            for (int t = T - 1; t >= tau; --t) {
                const double* rec = F + (size_t)t * L::S;
                if (lane < N) gv[lane] = rec[L::oD + lane] + pv[lane];
                else if (lane < N + M) {
                    int j = lane - N;
                    sv[j] = -hr * muv[j] * (wu[(size_t)t * M + j] - yu[(size_t)t * M + j]);
                }
                wave_sync();
                if (lane < N) {                      // p_t
                    double s = -(rec[L::oQx + lane] + hr * mxv[lane] * (wx[(size_t)t * N + lane] - yx[(size_t)t * N + lane]));
                    for (int l = 0; l < N; ++l) s += rec[L::oAcl + l * N + lane] * gv[l];
                    for (int j = 0; j < M; ++j) s += rec[L::oK + j * N + lane] * sv[j];
                    pv[lane] = s;
                } else if (lane < N + M) {           // k_t
                    int j = lane - N;
                    double s = 0.0;
                    for (int l = 0; l < N; ++l) s -= rec[L::oMinv + j * N + l] * gv[l];
                    for (int l = 0; l < M; ++l) s -= rec[L::oHinv + j * M + l] * sv[l];
                    kk[(size_t)t * M + j] = s;
                }
                wave_sync();
            }
            // forward sweep on the linear model: u = K x + k, x+ = Acl x + B k + c
            for (int t = tau; t < T; ++t) {
                const double* rec = F + (size_t)t * L::S;
                const double* xt = zx + (size_t)t * N;
                if (lane < N) {
                    double s = rec[L::oC + lane];
                    for (int l = 0; l < N; ++l) s += rec[L::oAcl + lane * N + l] * xt[l];
                    for (int j = 0; j < M; ++j) s += rec[L::oB + lane * M + j] * kk[(size_t)t * M + j];
                    zx[(size_t)(t + 1) * N + lane] = s;
                } else if (lane < N + M) {
                    int j = lane - N;
                    double s = kk[(size_t)t * M + j];
                    for (int l = 0; l < N; ++l) s += rec[L::oK + j * N + l] * xt[l];
                    zu[(size_t)t * M + j] = s;
                }
                wave_sync();
            }
            // projection + dual update (x_tau is fixed: only t > tau), residuals
            double rp = 0.0, rd = 0.0;
            for (int q = (tau + 1) * N + lane; q < (T + 1) * N; q += 64) {
                const int i = q % N, t = q / N;
                if (mxv[i] != 0.0) {
                    const double zr = al * zx[q] + (1.0 - al) * wx[q];
                    const double wn = fmin(fmax(zr + yx[q], zlo_(t, i)), zhi_(t, i));
                    rp = fmax(rp, fabs(zx[q] - wn));
                    rd = fmax(rd, fabs(wn - wx[q]));
                    yx[q] += zr - wn;
                    wx[q] = wn;
                }
            }
            for (int q = tau * M + lane; q < T * M; q += 64) {
                const int j = q % M, t = q / M;
                if (muv[j] != 0.0) {
                    const double zr = al * zu[q] + (1.0 - al) * wu[q];
                    const double wn = fmin(fmax(zr + yu[q], vlo_(t, j)), vhi_(t, j));
                    rp = fmax(rp, fabs(zu[q] - wn));
                    rd = fmax(rd, fabs(wn - wu[q]));
                    yu[q] += zr - wn;
                    wu[q] = wn;
                }
            }
            const double res = wave_max(fmax(rp, a.rho * rd));
            conv = res < a.eps;
            wave_sync();
        }
        it_max = max(it_max, it);
        n_fail += conv ? 0 : 1;
        if (a.single_tail) {
            // solve_tvlqr's return value: the plan of this one QP (xt_star (T+1,n), ut_star (T,m)) -- the
            // linear-model rollout of the converged iterate; for DU the controls are the u_prev blocks
            for (int q = lane; q < (T + 1) * NR; q += 64) a.x_new[q] = zx[(size_t)(q / NR) * N + q % NR];
            for (int q = lane; q < T * M; q += 64)
                a.u_new[q] = DU ? zx[(size_t)(q / M + 1) * N + NR + q % M] : zu[q];
            if (lane == 0) {
                a.info[0] = bad; a.info[1] = it_max; a.info[2] = n_fail;
            }
            return;
        }
        // first control of the tail solution (clipped), true dynamics step
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double v = fmin(fmax(zu[(size_t)tau * M + j], vlo_(tau, j)), vhi_(tau, j));
            if constexpr (DU) v = fmin(fmax(ub[j] + v, zlo_(tau + 1, NR + j)), zhi_(tau + 1, NR + j));
            ur[j] = v;
        }
        // running cost of the realised trajectory: IrsLqr.evaluate_cost (irs_lqr.py:121-137), or
        // for DU IrsLqrQuasistatic.eval_cost (irs_lqr_quasistatic.py:153-194: R on u_t - u_{t-1})
        {
            double e[NR], dv[M];
#pragma unroll
            for (int i = 0; i < NR; ++i) e[i] = xr[i] - a.xd[(size_t)tau * NR + i];
#pragma unroll
            for (int j = 0; j < M; ++j) dv[j] = DU ? ur[j] - (tau == 0 ? ub[j] : up[j]) : ur[j];
            cost += quad(a.Q, e, NR) + quad(a.R, dv, M);
        }
        Model::template step<double>(a.p, xr, ur, xn);
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
        wave_sync();
    }
    {
        double e[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) e[i] = xr[i] - a.xd[(size_t)T * NR + i];
        cost += quad(DU ? a.Qd : a.Q, e, NR);       // terminal: Qd (quasistatic :160-168) vs Q (irs_lqr.py:135-136)
    }
    if (lane == 0) {
        a.info[0] = bad; a.info[1] = it_max; a.info[2] = n_fail;
        if (a.cost) a.cost[0] = cost;
    }
}

template <class Model, bool DU>
int launch_box(const BoxArgs& a, hipStream_t st) {
    constexpr int N = Model::NX + (DU ? Model::NU : 0), M = Model::NU;
    const size_t bytes = BoxLayout<N, M>::doubles(a.T) * sizeof(double);
    if (bytes > 160 * 1024 - 512) {
        irs_set_error("irs_tvlqr_box_descent: horizon T=%d needs %zu bytes of LDS (max ~160 KB)", a.T, bytes);
        return IRS_ERR_UNSUPPORTED;
    }
    auto kern = box_descent_kernel<Model, DU>;
    static size_t granted = 0;           // per instantiation: the attribute call is a driver round trip (fused iterate)
    if (bytes > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            irs_set_error("irs_tvlqr_box_descent: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return IRS_ERR_HIP;
        }
        granted = bytes;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), bytes, st, a);
    return IRS_OK;
}

}  // namespace

extern "C" {

size_t irs_tvlqr_box_lds_bytes(int model, int T) {
    if (T <= 0) return 0;
    size_t r = 0;
    IRS_DISPATCH_MODEL(model, { r = BoxLayout<Model::NX, Model::NU>::doubles(T) * sizeof(double); });
    return r;
}

size_t irs_quasistatic_box_lds_bytes(int model, int T, int solver) {
    if (T <= 0) return 0;
    if (solver == 2) return irs_ctrlbox_lds_bytes(model, T);
    if (solver == 3) return irs_ctrlbox_mfma_lds_bytes(model, T);
    size_t r = 0;
    IRS_DISPATCH_MODEL(model, {
        if constexpr (has_u_into_x<Model>::value)
            r = BoxLayout<Model::NX + Model::NU, Model::NU>::doubles(T) * sizeof(double);
    });
    return r;
}

int irs_tvlqr_box_descent(int model, const double* params, int n_params, int T, const double* At,
                          const double* Bt, const double* ct, const double* Q, const double* Qd,
                          const double* R, double alpha_R, const double* xd_trj, const double* x0,
                          const double* xlo, const double* xhi, const double* ulo, const double* uhi,
                          double rho, double relax, int max_iter, double eps, double* x_new,
                          double* u_new, int* info, void* stream) {
    IRS_CHECK_ARG(T > 0 && At && Bt && ct && Q && Qd && R && xd_trj && x0 && xlo && xhi && ulo && uhi &&
                  x_new && u_new && info, "bad argument");
    IRS_CHECK_ARG(rho > 0.0 && relax > 0.0 && relax < 2.0 && max_iter > 0 && eps > 0.0, "bad ADMM parameter");
    return irs_tvlqr_box_descent_if(model, params, n_params, T, At, Bt, ct, Q, Qd, R, alpha_R, xd_trj, x0, xlo, xhi, ulo,
                                    uhi, rho, relax, max_iter, eps, x_new, u_new, nullptr, info, nullptr, stream);
}

int irs_tvlqr_box_descent_if(int model, const double* params, int n_params, int T, const double* At,
                             const double* Bt, const double* ct, const double* Q, const double* Qd,
                             const double* R, double alpha_R, const double* xd_trj, const double* x0,
                             const double* xlo, const double* xhi, const double* ulo, const double* uhi,
                             double rho, double relax, int max_iter, double eps, double* x_new,
                             double* u_new, double* cost, int* info, const int* run_flag, void* stream) {
    IRS_CHECK_ARG(T > 0 && At && Bt && ct && Q && Qd && R && xd_trj && x0 && xlo && xhi && ulo && uhi &&
                  x_new && u_new && info, "bad argument");
    IRS_CHECK_ARG(rho > 0.0 && relax > 0.0 && relax < 2.0 && max_iter > 0 && eps > 0.0, "bad ADMM parameter");
    BoxArgs a;
    a.act_io = nullptr;
    a.single_tail = 0;
    a.run_flag = nullptr;
    a.run_flag = run_flag;
    int rc = irs_load_params(model, params, n_params, &a.p);
    if (rc != IRS_OK) return rc;
    a.At = At; a.Bt = Bt; a.ct = ct; a.Q = Q; a.Qd = Qd; a.R = R; a.xd = xd_trj; a.x0 = x0;
    a.xlo = xlo; a.xhi = xhi; a.ulo = ulo; a.uhi = uhi; a.dlo = nullptr; a.dhi = nullptr;
    a.sx = 0; a.su = 0; a.sd = 0;
    a.x_new = x_new; a.u_new = u_new; a.cost = cost; a.info = info;
    a.alpha = alpha_R; a.rho = rho; a.relax = relax; a.eps = eps; a.T = T; a.max_iter = max_iter;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, { rc = launch_box<Model, false>(a, st); });
    if (rc != IRS_OK) return rc;
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_quasistatic_box_descent(int model, const double* params, int n_params, int T, const double* At,
                                const double* Bt, const double* ct, const double* Q, const double* Qd,
                                const double* R, const double* xd_trj, const double* x0,
                                const double* x_lo, const double* x_hi, const double* u_lo,
                                const double* u_hi, const double* du_lo, const double* du_hi,
                                int solver, double rho, double relax, int max_iter, double eps,
                                double* x_new, double* u_new, double* cost, int* info, void* stream) {
    return irs_quasistatic_box_descent_ws(model, params, n_params, T, At, Bt, ct, Q, Qd, R, xd_trj, x0, x_lo, x_hi,
                                          u_lo, u_hi, du_lo, du_hi, solver, rho, relax, max_iter, eps, x_new,
                                          u_new, cost, info, nullptr, stream);
}

int irs_quasistatic_box_descent_ws(int model, const double* params, int n_params, int T, const double* At,
                                   const double* Bt, const double* ct, const double* Q, const double* Qd,
                                   const double* R, const double* xd_trj, const double* x0,
                                   const double* x_lo, const double* x_hi, const double* u_lo,
                                   const double* u_hi, const double* du_lo, const double* du_hi,
                                   int solver, double rho, double relax, int max_iter, double eps,
                                   double* x_new, double* u_new, double* cost, int* info, double* act_io,
                                   void* stream) {
    return irs_quasistatic_box_descent_wsx(model, params, n_params, T, At, Bt, ct, Q, Qd, R, xd_trj, x0, x_lo, x_hi,
                                           u_lo, u_hi, du_lo, du_hi, solver, rho, relax, max_iter, eps, x_new,
                                           u_new, cost, info, act_io, nullptr, 0, stream);
}

size_t irs_quasistatic_descent_workspace_bytes(int model, int T, int solver) {
    if (T <= 0 || (solver != 0 && solver != 3)) return 0;
    const size_t lds = irs_ctrlbox_mfma_lds_bytes(model, T);
    return (lds == 0 || lds <= (size_t)(160 * 1024 - 512)) ? 0 : irs_ctrlbox_mfma_record_bytes(model, T);
}

int irs_quasistatic_box_descent_wsx(int model, const double* params, int n_params, int T, const double* At,
                                    const double* Bt, const double* ct, const double* Q, const double* Qd,
                                    const double* R, const double* xd_trj, const double* x0,
                                    const double* x_lo, const double* x_hi, const double* u_lo,
                                    const double* u_hi, const double* du_lo, const double* du_hi,
                                    int solver, double rho, double relax, int max_iter, double eps,
                                    double* x_new, double* u_new, double* cost, int* info, double* act_io,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    IRS_CHECK_ARG(T > 0 && At && Bt && ct && Q && Qd && R && xd_trj && x0 && x_new && u_new && info, "bad argument");
    IRS_CHECK_ARG(solver >= 0 && solver <= 3,
                  "solver must be 0 (auto), 1 (ADMM), 2 (active set, lanes) or 3 (active set, matrix-core tiles)");
    IRS_CHECK_ARG((x_lo == nullptr) == (x_hi == nullptr) && (u_lo == nullptr) == (u_hi == nullptr) &&
                  (du_lo == nullptr) == (du_hi == nullptr), "give both sides of a bound or neither");
    IRS_CHECK_ARG(rho > 0.0 && relax > 0.0 && relax < 2.0 && max_iter > 0 && eps > 0.0, "bad ADMM parameter");
    BoxArgs a;
    a.act_io = act_io;
    a.single_tail = 0;
    a.run_flag = nullptr;
    int rc = irs_load_params(model, params, n_params, &a.p);
    if (rc != IRS_OK) return rc;
    int n, m, np;
    irs_model_info(model, &n, &m, &np);
    a.At = At; a.Bt = Bt; a.ct = ct; a.Q = Q; a.Qd = Qd; a.R = R; a.xd = xd_trj; a.x0 = x0;
    a.xlo = x_lo; a.xhi = x_hi; a.ulo = u_lo; a.uhi = u_hi; a.dlo = du_lo; a.dhi = du_hi;
    a.sx = n; a.su = m; a.sd = m;
    a.x_new = x_new; a.u_new = u_new; a.cost = cost; a.info = info;
    a.alpha = 1.0;      // tv_lqr.py:107 adds du'R du as an expression: the full quadratic
    a.rho = rho; a.relax = relax; a.eps = eps; a.T = T; a.max_iter = max_iter;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // one control box (or none) and no state bounds: the exact active-set solvers apply
    const bool one_box = x_lo == nullptr && !(u_lo != nullptr && du_lo != nullptr);
    if ((solver == 2 || solver == 3) && !one_box) {
        irs_set_error("irs_quasistatic_box_descent: the active-set solvers handle ONE of u / du bounds and no x bounds");
        return IRS_ERR_UNSUPPORTED;
    }
    const size_t kLds = (size_t)(160 * 1024 - 512);
    const int kind = du_lo != nullptr ? 1 : 0;
    if (one_box && (solver == 3 || solver == 0)) {
        // matrix-core tiles: records on chip when they fit, in the caller's workspace otherwise
        const size_t lds = irs_ctrlbox_mfma_lds_bytes(model, T);
        const bool fits = lds != 0 && (lds <= kLds ||
                                       (workspace != nullptr && workspace_bytes >= irs_ctrlbox_mfma_record_bytes(model, T)));
        if (solver == 3 || fits) {
            rc = irs_ctrlbox_mfma_launch(model, a, kind, static_cast<double*>(workspace), workspace_bytes, st);
            if (rc != IRS_OK) return rc;
            IRS_CHECK_LAUNCH();
            return IRS_OK;
        }
    }
    if (solver == 2 || (solver == 0 && one_box && irs_ctrlbox_lds_bytes(model, T) <= kLds)) {
        rc = irs_ctrlbox_launch(model, a, kind, st);
        if (rc != IRS_OK) return rc;
        IRS_CHECK_LAUNCH();
        return IRS_OK;
    }
    rc = IRS_ERR_UNSUPPORTED;
    IRS_DISPATCH_MODEL(model, {
        if constexpr (has_u_into_x<Model>::value) rc = launch_box<Model, true>(a, st);
        else irs_set_error("irs_quasistatic_box_descent: model %d is not position controlled", model);
    });
    if (rc != IRS_OK) return rc;
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

// solve_tvlqr (irs_lqr/tv_lqr.py:30-145) stand-alone: ONE bounded QP, its plan returned.
int irs_tvlqr_box_solve(int model, const double* params, int n_params, int T, const double* At, const double* Bt,
                        const double* ct, const double* Q, const double* Qd, const double* R, double alpha_R,
                        const double* xd_trj, const double* x0, int position_controlled,
                        const double* x_lo, const double* x_hi, const double* u_lo, const double* u_hi,
                        const double* du_lo, const double* du_hi, double rho, double relax, int max_iter,
                        double eps, double* x_star, double* u_star, int* info, void* stream) {
    IRS_CHECK_ARG(T > 0 && At && Bt && ct && Q && Qd && R && xd_trj && x0 && x_star && u_star && info, "bad argument");
    IRS_CHECK_ARG((x_lo == nullptr) == (x_hi == nullptr) && (u_lo == nullptr) == (u_hi == nullptr) &&
                  (du_lo == nullptr) == (du_hi == nullptr), "give both sides of a bound or neither");
    IRS_CHECK_ARG(position_controlled || du_lo == nullptr, "du bounds need the position-controlled form");
    IRS_CHECK_ARG(rho > 0.0 && relax > 0.0 && relax < 2.0 && max_iter > 0 && eps > 0.0, "bad ADMM parameter");
    BoxArgs a;
    a.act_io = nullptr;
    a.single_tail = 1;
    a.run_flag = nullptr;
    int rc = irs_load_params(model, params, n_params, &a.p);
    if (rc != IRS_OK) return rc;
    int n, m, np;
    irs_model_info(model, &n, &m, &np);
    a.At = At; a.Bt = Bt; a.ct = ct; a.Q = Q; a.Qd = Qd; a.R = R; a.xd = xd_trj; a.x0 = x0;
    a.xlo = x_lo; a.xhi = x_hi; a.ulo = u_lo; a.uhi = u_hi; a.dlo = du_lo; a.dhi = du_hi;
    a.sx = n; a.su = m; a.sd = m;
    a.x_new = x_star; a.u_new = u_star; a.cost = nullptr; a.info = info;
    a.alpha = alpha_R; a.rho = rho; a.relax = relax; a.eps = eps; a.T = T; a.max_iter = max_iter;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = IRS_ERR_UNSUPPORTED;
    if (position_controlled) {
        IRS_DISPATCH_MODEL(model, {
            if constexpr (has_u_into_x<Model>::value) rc = launch_box<Model, true>(a, st);
            else irs_set_error("irs_tvlqr_box_solve: model %d is not position controlled", model);
        });
    } else {
        IRS_DISPATCH_MODEL(model, { rc = launch_box<Model, false>(a, st); });
    }
    if (rc != IRS_OK) return rc;
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

}  // extern "C"
