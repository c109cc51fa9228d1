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
// One wave, f64, everything in (dynamic) LDS: a latency-bound chain like the Riccati pass.
#include "irs_common.hpp"

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

struct BoxArgs {
    ModelParams p;
    const double *At, *Bt, *ct, *Q, *Qd, *R, *xd, *x0;
    const double *xlo, *xhi, *ulo, *uhi;     // (n), (n), (m), (m); +-inf = unbounded
    double *x_new, *u_new, *cost;
    int* info;                               // [0] Hessian not PD at t+1, [1] max ADMM iterations used,
                                             // [2] number of tail problems that hit max_iter
    double alpha, rho, relax, eps;
    int T, max_iter;
};

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

template <class Model>
__global__ __launch_bounds__(64) void box_descent_kernel(BoxArgs a) {
    constexpr int N = Model::NX, M = Model::NU;
    using L = BoxLayout<N, M>;
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
    double* mxv = sv + N;                              // bounded masks / bounds
    double* xlo = mxv + N;
    double* xhi = xlo + N;
    double* muv = xhi + N;                             // (M) each
    double* ulo = muv + M;
    double* uhi = ulo + M;
    double* Hs = uhi + M;                              // M x M scratch (<= 16)

    // ---- setup ------------------------------------------------------------------------
    if (lane < N) {
        xlo[lane] = a.xlo[lane];
        xhi[lane] = a.xhi[lane];
        mxv[lane] = (isfinite(a.xlo[lane]) || isfinite(a.xhi[lane])) ? 1.0 : 0.0;
    }
    if (lane < M) {
        ulo[lane] = a.ulo[lane];
        uhi[lane] = a.uhi[lane];
        muv[lane] = (isfinite(a.ulo[lane]) || isfinite(a.uhi[lane])) ? 1.0 : 0.0;
    }
    for (int q = lane; q < (T + 1) * N; q += 64) { wx[q] = 0.0; yx[q] = 0.0; zx[q] = 0.0; }
    for (int q = lane; q < T * M; q += 64) { wu[q] = 0.0; yu[q] = 0.0; zu[q] = 0.0; kk[q] = 0.0; }
    wave_sync();
    const double hr = 0.5 * a.rho;
    auto qs = [&](const double* Qm, int i, int j) { return 0.5 * (Qm[i * N + j] + Qm[j * N + i]); };
    // P_T = Qd + rho/2 Mx ; qx_T = Qd xd_T
    for (int q = lane; q < N * N; q += 64) {
        int i = q / N, j = q % N;
        P[q] = qs(a.Qd, i, j) + (i == j ? hr * mxv[i] : 0.0);
    }
    if (lane < N) {
        double s = 0.0;
        for (int j = 0; j < N; ++j) s += qs(a.Qd, lane, j) * a.xd[(size_t)T * N + j];
        qxT[lane] = s;
    }
    wave_sync();

    // ---- factorisation: backward Riccati with Q^ = Q + rho/2 Mx, R^ = alpha R + rho/2 Mu --
    int bad = 0;
    for (int t = T - 1; t >= 0; --t) {
        double* rec = F + (size_t)t * L::S;
        for (int q = lane; q < N * N; q += 64) Am[q] = a.At[(size_t)t * N * N + q];
        for (int q = lane; q < N * M; q += 64) rec[L::oB + q] = a.Bt[(size_t)t * N * M + q];
        if (lane < N) {
            rec[L::oC + lane] = a.ct[(size_t)t * N + lane];
            double s = 0.0;
            for (int j = 0; j < N; ++j) s += qs(a.Q, lane, j) * a.xd[(size_t)t * N + j];
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
    double xr[N], ur[M], xn[N];
#pragma unroll
    for (int i = 0; i < N; ++i) xr[i] = a.x0[i];
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) a.x_new[i] = xr[i];
    }
    int it_max = 0, n_fail = 0;
    const double al = a.relax;
    for (int tau = 0; tau < T; ++tau) {
        if (lane < N) {
            double v = xr[0];
#pragma unroll
            for (int i = 1; i < N; ++i) v = (i == lane) ? xr[i] : v;
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
                const int i = q % N;
                if (mxv[i] != 0.0) {
                    const double zr = al * zx[q] + (1.0 - al) * wx[q];
                    const double wn = fmin(fmax(zr + yx[q], xlo[i]), xhi[i]);
                    rp = fmax(rp, fabs(zx[q] - wn));
                    rd = fmax(rd, fabs(wn - wx[q]));
                    yx[q] += zr - wn;
                    wx[q] = wn;
                }
            }
            for (int q = tau * M + lane; q < T * M; q += 64) {
                const int j = q % M;
                if (muv[j] != 0.0) {
                    const double zr = al * zu[q] + (1.0 - al) * wu[q];
                    const double wn = fmin(fmax(zr + yu[q], ulo[j]), uhi[j]);
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
        // first control of the tail solution (clipped), true dynamics step
#pragma unroll
        for (int j = 0; j < M; ++j) ur[j] = fmin(fmax(zu[(size_t)tau * M + j], ulo[j]), uhi[j]);
        Model::template step<double>(a.p, xr, ur, xn);
#pragma unroll
        for (int i = 0; i < N; ++i) xr[i] = xn[i];
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < M; ++j) a.u_new[(size_t)tau * M + j] = ur[j];
#pragma unroll
            for (int i = 0; i < N; ++i) a.x_new[(size_t)(tau + 1) * N + i] = xr[i];
        }
        wave_sync();
    }
    if (lane == 0) { a.info[0] = bad; a.info[1] = it_max; a.info[2] = n_fail; }
}

template <class Model>
int launch_box(const BoxArgs& a, hipStream_t st) {
    constexpr int N = Model::NX, M = Model::NU;
    const size_t bytes = BoxLayout<N, M>::doubles(a.T) * sizeof(double);
    if (bytes > 160 * 1024 - 512) {
        irs_set_error("irs_tvlqr_box_descent: horizon T=%d needs %zu bytes of LDS (max ~160 KB)", a.T, bytes);
        return IRS_ERR_UNSUPPORTED;
    }
    auto kern = box_descent_kernel<Model>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        irs_set_error("irs_tvlqr_box_descent: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return IRS_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), bytes, st, a);
    return IRS_OK;
}

}  // namespace

extern "C" {

size_t irs_tvlqr_box_lds_bytes(int model, int T) {
    if (T <= 0) return 0;
    switch (model) {
        case IRS_MODEL_PENDULUM: return BoxLayout<2, 1>::doubles(T) * sizeof(double);
        case IRS_MODEL_QUADROTOR: return BoxLayout<12, 4>::doubles(T) * sizeof(double);
        case IRS_MODEL_BICYCLE: return BoxLayout<5, 2>::doubles(T) * sizeof(double);
        case IRS_MODEL_THREE_CART: return BoxLayout<6, 2>::doubles(T) * sizeof(double);
        case IRS_MODEL_PLANAR_HAND: return BoxLayout<7, 4>::doubles(T) * sizeof(double);
    }
    return 0;
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
    BoxArgs a;
    int rc = irs_load_params(model, params, n_params, &a.p);
    if (rc != IRS_OK) return rc;
    a.At = At; a.Bt = Bt; a.ct = ct; a.Q = Q; a.Qd = Qd; a.R = R; a.xd = xd_trj; a.x0 = x0;
    a.xlo = xlo; a.xhi = xhi; a.ulo = ulo; a.uhi = uhi;
    a.x_new = x_new; a.u_new = u_new; a.cost = nullptr; a.info = info;
    a.alpha = alpha_R; a.rho = rho; a.relax = relax; a.eps = eps; a.T = T; a.max_iter = max_iter;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, { rc = launch_box<Model>(a, st); });
    if (rc != IRS_OK) return rc;
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

}  // extern "C"
