// TV-LQR (irs_lqr/tv_lqr.py:30-145 with inactive bounds) as a backward Riccati pass,
// and the forward loop of IrsLqr.local_descent (irs_lqr/irs_lqr.py:169-184).
//
// Both are T-step sequential chains on matrices of order n <= 32: latency-bound, not
// bandwidth- or FLOP-bound.  Each runs as ONE wave (64 lanes) with its state in LDS
// (f64) -- no inter-workgroup synchronisation, no host round trips; lanes split the
// output elements of every small product.
#include "irs_common.hpp"

namespace {

constexpr int kMaxN = 32;
constexpr int kMaxM = 16;

struct RiccatiArgs {
    const double* At; const double* Bt; const double* ct;
    const double* Q; const double* Qd; const double* R;
    const double* xd;
    double* K; double* k;
    int* info;
    double alpha;
    int n, m, T;
};

// LDS matrices are stored with leading dimension n (or m) -- runtime sizes.
__global__ __launch_bounds__(64) void riccati_kernel(RiccatiArgs a) {
    const int n = a.n, m = a.m, T = a.T, lane = threadIdx.x;
    __shared__ double P[kMaxN * kMaxN];     // value Hessian (n x n)
    __shared__ double pv[kMaxN];            // value gradient (n)
    __shared__ double A[kMaxN * kMaxN];
    __shared__ double B[kMaxN * kMaxM];
    __shared__ double PB[kMaxN * kMaxM];    // P B          (n x m)
    __shared__ double W[kMaxN * kMaxN];     // scratch      (n x n)
    __shared__ double Acl[kMaxN * kMaxN];   // A + B K      (n x n)
    __shared__ double Hm[kMaxM * (kMaxM + 1)];  // alpha R + B'PB (m x m), ld m+1
    __shared__ double Kt[kMaxM * kMaxN];    // gain         (m x n)
    __shared__ double qv[kMaxN];            // P c + p
    __shared__ double kt[kMaxM];
    __shared__ double Qs[kMaxN * kMaxN];
    __shared__ double Rs[kMaxM * kMaxM];
    __shared__ int bad;
    const int ldh = m + 1;

    for (int q = lane; q < n * n; q += 64) { P[q] = a.Qd[q]; Qs[q] = a.Q[q]; }
    for (int q = lane; q < m * m; q += 64) Rs[q] = a.R[q];
    if (lane == 0) bad = 0;
    __syncthreads();
    if (lane < n) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s -= P[lane * n + j] * a.xd[(size_t)T * n + j];
        pv[lane] = s;
    }
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const double* At = a.At + (size_t)t * n * n;
        const double* Bt = a.Bt + (size_t)t * n * m;
        const double* ct = a.ct + (size_t)t * n;
        for (int q = lane; q < n * n; q += 64) A[q] = At[q];
        for (int q = lane; q < n * m; q += 64) B[q] = Bt[q];
        __syncthreads();
        // PB = P B ; q = P c + p
        for (int q = lane; q < n * m; q += 64) {
            int i = q / m, j = q % m;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += P[i * n + l] * B[l * m + j];
            PB[q] = s;
        }
        if (lane < n) {
            double s = pv[lane];
            for (int l = 0; l < n; ++l) s += P[lane * n + l] * ct[l];
            qv[lane] = s;
        }
        __syncthreads();
        // H = alpha R + B' PB ; G1 = PB' A (into Kt) ; g = B' q (into kt)
        for (int q = lane; q < m * m; q += 64) {
            int i = q / m, j = q % m;
            double s = a.alpha * Rs[q];
            for (int l = 0; l < n; ++l) s += B[l * m + i] * PB[l * m + j];
            Hm[i * ldh + j] = s;
        }
        for (int q = lane; q < m * n; q += 64) {
            int i = q / n, j = q % n;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += PB[l * m + i] * A[l * n + j];
            Kt[q] = s;
        }
        if (lane < m) {
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += B[l * m + lane] * qv[l];
            kt[lane] = s;
        }
        __syncthreads();
        // Cholesky of H (m x m), lower in place
        for (int j = 0; j < m; ++j) {
            double djj = Hm[j * ldh + j];
            if (!(djj > 0.0)) {
                if (lane == 0 && bad == 0) bad = t + 1;
                djj = 1.0;
            }
            double l = sqrt(djj);
            __syncthreads();
            if (lane == j) Hm[j * ldh + j] = l;
            if (lane > j && lane < m) Hm[lane * ldh + j] /= l;
            __syncthreads();
            for (int q = lane; q < m * m; q += 64) {
                int r = q / m, c = q % m;
                if (c > j && r >= c) Hm[r * ldh + c] -= Hm[r * ldh + j] * Hm[c * ldh + j];
            }
            __syncthreads();
        }
        // K = -H^-1 G1 (one lane per column of G1), k = -H^-1 g (lane n)
        if (lane <= n) {
            double y[kMaxM];
            for (int i = 0; i < m; ++i) {
                double s = (lane < n) ? Kt[i * n + lane] : kt[i];
                for (int l = 0; l < i; ++l) s -= Hm[i * ldh + l] * y[l];
                y[i] = s / Hm[i * ldh + i];
            }
            for (int i = m - 1; i >= 0; --i) {
                double s = y[i];
                for (int l = i + 1; l < m; ++l) s -= Hm[l * ldh + i] * y[l];
                y[i] = s / Hm[i * ldh + i];
            }
            for (int i = 0; i < m; ++i) {
                if (lane < n) Kt[i * n + lane] = -y[i];
                else kt[i] = -y[i];
            }
        }
        __syncthreads();
        for (int q = lane; q < m * n; q += 64) a.K[(size_t)t * m * n + q] = Kt[q];
        if (lane < m) a.k[(size_t)t * m + lane] = kt[lane];
        // Acl = A + B K
        for (int q = lane; q < n * n; q += 64) {
            int i = q / n, j = q % n;
            double s = A[q];
            for (int l = 0; l < m; ++l) s += B[i * m + l] * Kt[l * n + j];
            Acl[q] = s;
        }
        __syncthreads();
        // W = P Acl ; p_new = -Q xd_t + Acl' q
        for (int q = lane; q < n * n; q += 64) {
            int i = q / n, j = q % n;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += P[i * n + l] * Acl[l * n + j];
            W[q] = s;
        }
        double pnew = 0.0;
        if (lane < n) {
            const double* xd = a.xd + (size_t)t * n;
            for (int l = 0; l < n; ++l) pnew += Acl[l * n + lane] * qv[l] - Qs[lane * n + l] * xd[l];
        }
        __syncthreads();
        // P = Q + A' W, symmetrised
        for (int q = lane; q < n * n; q += 64) {
            int i = q / n, j = q % n;
            double s = 0.0, s2 = 0.0;
            for (int l = 0; l < n; ++l) {
                s += A[l * n + i] * W[l * n + j];
                s2 += A[l * n + j] * W[l * n + i];
            }
            P[q] = Qs[q] + 0.5 * (s + s2);
        }
        if (lane < n) pv[lane] = pnew;
        __syncthreads();
    }
    if (lane == 0) a.info[0] = bad;
}

__global__ __launch_bounds__(64) void linear_rollout_kernel(int n, int m, int T, const double* At,
                                                            const double* Bt, const double* ct,
                                                            const double* K, const double* k,
                                                            const double* x0, double* xs, double* us) {
    __shared__ double x[kMaxN];
    __shared__ double u[kMaxM];
    const int lane = threadIdx.x;
    if (lane < n) { x[lane] = x0[lane]; xs[lane] = x0[lane]; }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (lane < m) {
            double s = k[(size_t)t * m + lane];
            for (int j = 0; j < n; ++j) s += K[((size_t)t * m + lane) * n + j] * x[j];
            u[lane] = s;
            us[(size_t)t * m + lane] = s;
        }
        __syncthreads();
        double xn = 0.0;
        if (lane < n) {
            xn = ct[(size_t)t * n + lane];
            for (int j = 0; j < n; ++j) xn += At[((size_t)t * n + lane) * n + j] * x[j];
            for (int j = 0; j < m; ++j) xn += Bt[((size_t)t * n + lane) * m + j] * u[j];
        }
        __syncthreads();
        if (lane < n) { x[lane] = xn; xs[(size_t)(t + 1) * n + lane] = xn; }
        __syncthreads();
    }
}

// cost of one stage: (x-xd)'Q(x-xd) [+ u'Ru]
template <int n>
__device__ __forceinline__ double quad_err(const double* Q, const double* x, const double* xd) {
    double e[n], s = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) e[i] = x[i] - xd[i];
#pragma unroll
    for (int i = 0; i < n; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < n; ++j) r += Q[i * n + j] * e[j];
        s += e[i] * r;
    }
    return s;
}

// Closed-loop (K != null) or open-loop (K == null, u = u_in) rollout on the TRUE
// dynamics + evaluate_cost.  Sequential: every lane of the single wave carries the
// same state in registers (no divergence, no broadcasts); lane 0 stores.
template <class Model>
__global__ __launch_bounds__(64) void rollout_kernel(ModelParams p, int T, const double* K,
                                                     const double* k, const double* u_in,
                                                     const double* x0, const double* Q,
                                                     const double* R, const double* xd_trj,
                                                     double* x_out, double* u_out, double* cost_out) {
    constexpr int n = Model::NX, m = Model::NU;
    __shared__ double Qs[n * n];
    __shared__ double Rs[m * m];
    const int lane = threadIdx.x;
    for (int q = lane; q < n * n; q += 64) Qs[q] = Q[q];
    for (int q = lane; q < m * m; q += 64) Rs[q] = R[q];
    __syncthreads();
    double x[n], u[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = x0[i];
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < n; ++i) x_out[i] = x[i];
    }
    double cost = 0.0;
    for (int t = 0; t < T; ++t) {
        if (K != nullptr) {
#pragma unroll
            for (int i = 0; i < m; ++i) {
                double s = k[(size_t)t * m + i];
#pragma unroll
                for (int j = 0; j < n; ++j) s += K[((size_t)t * m + i) * n + j] * x[j];
                u[i] = s;
            }
        } else {
#pragma unroll
            for (int i = 0; i < m; ++i) u[i] = u_in[(size_t)t * m + i];
        }
        cost += quad_err<n>(Qs, x, xd_trj + (size_t)t * n);
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) r += Rs[i * m + j] * u[j];
            cost += u[i] * r;
        }
        Model::template step<double>(p, x, u, xn);
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xn[i];
        if (lane == 0) {
            if (u_out != nullptr) {
#pragma unroll
                for (int i = 0; i < m; ++i) u_out[(size_t)t * m + i] = u[i];
            }
#pragma unroll
            for (int i = 0; i < n; ++i) x_out[(size_t)(t + 1) * n + i] = x[i];
        }
    }
    // terminal term uses Q, not Qd (irs_lqr/irs_lqr.py:135-136)
    cost += quad_err<n>(Qs, x, xd_trj + (size_t)T * n);
    if (lane == 0) cost_out[0] = cost;
}

// evaluate_cost of a given trajectory pair: lanes stride over t, f64 wave reduction.
__global__ __launch_bounds__(64) void evaluate_cost_kernel(int n, int m, int T, const double* x_trj,
                                                           const double* u_trj, const double* Q,
                                                           const double* R, const double* xd_trj,
                                                           double* cost_out) {
    const int lane = threadIdx.x;
    double acc = 0.0;
    for (int t = lane; t <= T; t += 64) {
        const double* x = x_trj + (size_t)t * n;
        const double* xd = xd_trj + (size_t)t * n;
        for (int i = 0; i < n; ++i) {
            double r = 0.0;
            for (int j = 0; j < n; ++j) r += Q[i * n + j] * (x[j] - xd[j]);
            acc += (x[i] - xd[i]) * r;
        }
        if (t < T) {
            const double* u = u_trj + (size_t)t * m;
            for (int i = 0; i < m; ++i) {
                double r = 0.0;
                for (int j = 0; j < m; ++j) r += R[i * m + j] * u[j];
                acc += u[i] * r;
            }
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (lane == 0) cost_out[0] = acc;
}

}  // namespace

extern "C" {

int irs_evaluate_cost(int n, int m, int T, const double* x_trj, const double* u_trj, const double* Q,
                      const double* R, const double* xd_trj, double* cost, void* stream) {
    IRS_CHECK_ARG(n > 0 && n <= kMaxN && m > 0 && m <= kMaxM && T > 0, "need 0<n<=32, 0<m<=16, T>0");
    IRS_CHECK_ARG(x_trj && u_trj && Q && R && xd_trj && cost, "null pointer");
    hipLaunchKernelGGL(evaluate_cost_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), n, m,
                       T, x_trj, u_trj, Q, R, xd_trj, cost);
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_tvlqr_riccati(int n, int m, int T, const double* At, const double* Bt, const double* ct,
                      const double* Q, const double* Qd, const double* R, double alpha_R,
                      const double* xd_trj, double* K, double* k, int* info, void* stream) {
    IRS_CHECK_ARG(n > 0 && n <= kMaxN && m > 0 && m <= kMaxM && T > 0, "need 0<n<=32, 0<m<=16, T>0");
    IRS_CHECK_ARG(At && Bt && ct && Q && Qd && R && xd_trj && K && k && info, "null pointer");
    RiccatiArgs a{At, Bt, ct, Q, Qd, R, xd_trj, K, k, info, alpha_R, n, m, T};
    hipLaunchKernelGGL(riccati_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_tvlqr_linear_rollout(int n, int m, int T, const double* At, const double* Bt,
                             const double* ct, const double* K, const double* k, const double* x0,
                             double* x_star, double* u_star, void* stream) {
    IRS_CHECK_ARG(n > 0 && n <= kMaxN && m > 0 && m <= kMaxM && T > 0, "need 0<n<=32, 0<m<=16, T>0");
    IRS_CHECK_ARG(At && Bt && ct && K && k && x0 && x_star && u_star, "null pointer");
    hipLaunchKernelGGL(linear_rollout_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       n, m, T, At, Bt, ct, K, k, x0, x_star, u_star);
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_closed_loop_rollout(int model, const double* params, int n_params, int T, const double* K,
                            const double* k, const double* x0, const double* Q, const double* R,
                            const double* xd_trj, double* x_new, double* u_new, double* cost,
                            void* stream) {
    IRS_CHECK_ARG(T > 0 && K && k && x0 && Q && R && xd_trj && x_new && u_new && cost, "bad argument");
    ModelParams p;
    int rc = irs_load_params(model, params, n_params, &p);
    if (rc != IRS_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, {
        hipLaunchKernelGGL((rollout_kernel<Model>), dim3(1), dim3(64), 0, st, p, T, K, k,
                           (const double*)nullptr, x0, Q, R, xd_trj, x_new, u_new, cost);
    });
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_rollout_cost(int model, const double* params, int n_params, int T, const double* x0,
                     const double* u_trj, const double* Q, const double* R, const double* xd_trj,
                     double* x_trj, double* cost, void* stream) {
    IRS_CHECK_ARG(T > 0 && x0 && u_trj && Q && R && xd_trj && x_trj && cost, "bad argument");
    ModelParams p;
    int rc = irs_load_params(model, params, n_params, &p);
    if (rc != IRS_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, {
        hipLaunchKernelGGL((rollout_kernel<Model>), dim3(1), dim3(64), 0, st, p, T,
                           (const double*)nullptr, (const double*)nullptr, u_trj, x0, Q, R, xd_trj,
                           x_trj, (double*)nullptr, cost);
    });
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

}  // extern "C"
