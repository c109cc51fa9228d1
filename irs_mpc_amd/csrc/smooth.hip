// Randomised-smoothing linearisation: get_TV_matrices of
//   irs_lqr/irs_lqr_zero_order.py:38-63   (ZERO_ORDER_AB)
//   irs_lqr/irs_lqr_first_order.py:28-54  (FIRST_ORDER)
//   irs_lqr/quasistatic_dynamics.py:242-266 (ZERO_ORDER_B, u-only noise)
//   irs_lqr/irs_lqr_exact.py:15-31        (exact)
// as a streaming map-reduce over the (T x N) grid of independent one-step samples.
//
//   pass 1  smooth_accum_kernel   grid (nblk, T) x 256 threads.  Each lane streams
//           its samples' z=[dx|du] (f32, vector loads, consecutive lanes read
//           consecutive records), evaluates the model functor in f32, accumulates
//           the P sufficient statistics in registers, then the workgroup reduces
//           them with wave shuffles + one LDS hop -> partial[t][blk][P] (f32).
//   pass 2  reduce_partials_kernel  partial -> sums[t][P] in f64, fixed order.
//           (`sums` is what several GPUs all-reduce.)
//   pass 3  smooth_finalize_kernel  one wave per timestep: Jacobi-scaled Cholesky of
//           the Gram matrix in f64 -> A_t, B_t;  c_t = f(x_t,u_t) - A_t x_t - B_t u_t.
#include "irs_common.hpp"
#include "philox.hpp"
#include "reduce.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

template <class Model, int MODE>
struct SmoothTraits {
    static constexpr int n = Model::NX, m = Model::NU, d = n + m;
    // perturbed components that enter the least-squares design matrix
    static constexpr int NZ = (MODE == IRS_SMOOTH_ZERO_ORDER_B) ? m : d;
    static constexpr int Z0 = (MODE == IRS_SMOOTH_ZERO_ORDER_B) ? n : 0;  // first one
    static constexpr int NG = NZ * (NZ + 1) / 2;
    static constexpr int P = (MODE == IRS_SMOOTH_FIRST_ORDER) ? n * d : NG + NZ * n;
    static constexpr int PP = irs_reduce_pad(P);
};

struct SmoothArgs {
    ModelParams p;
    const double* x_trj;
    const double* u_trj;
    const float* dx;
    const float* du;
    float std[32];
    unsigned long long seed;
    unsigned long long sample_offset;
    unsigned int iter;
    int T, N, chunk, nblk;
    float* partial;
};

template <int K>
__device__ __forceinline__ void load_row(const float* __restrict__ ptr, float* out) {
    if constexpr (K % 4 == 0) {
#pragma unroll
        for (int i = 0; i < K / 4; ++i) {
            float4 v = reinterpret_cast<const float4*>(ptr)[i];
            out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
        }
    } else if constexpr (K % 2 == 0) {
#pragma unroll
        for (int i = 0; i < K / 2; ++i) {
            float2 v = reinterpret_cast<const float2*>(ptr)[i];
            out[2 * i] = v.x; out[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) out[i] = ptr[i];
    }
}

template <class Model, int MODE, bool RNG>
__global__ __launch_bounds__(kBlock) void smooth_accum_kernel(SmoothArgs a) {
    using TR = SmoothTraits<Model, MODE>;
    constexpr int n = TR::n, m = TR::m, d = TR::d, NZ = TR::NZ, Z0 = TR::Z0;
    __shared__ float red[kWaves * TR::PP];

    const int t = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;

    // nominal point of this timestep (every lane keeps its own copy in registers)
    float xb[n], ub[m], f0[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = (float)a.x_trj[(size_t)t * n + i];
#pragma unroll
    for (int j = 0; j < m; ++j) ub[j] = (float)a.u_trj[(size_t)t * m + j];
    if constexpr (MODE != IRS_SMOOTH_FIRST_ORDER) Model::template step<float>(a.p, xb, ub, f0);

    float acc[TR::PP];
#pragma unroll
    for (int i = 0; i < TR::PP; ++i) acc[i] = 0.f;

    const int s_end = min(a.N, (blk + 1) * a.chunk);
    for (int s = blk * a.chunk + tid; s < s_end; s += kBlock) {
        float z[d];
        if constexpr (RNG) {
            constexpr int j0 = Z0 / 4;
            const unsigned long long gidx = a.sample_offset + (unsigned long long)s;
#pragma unroll
            for (int j = j0; j < (d + 3) / 4; ++j) {
                float g[4];
                philox_normal4(gidx, (unsigned)t, (unsigned)j, a.iter, a.seed, g);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * j + c < d) z[4 * j + c] = g[c] * a.std[4 * j + c];
            }
#pragma unroll
            for (int i = 0; i < Z0; ++i) z[i] = 0.f;
        } else {
            const size_t row = (size_t)t * a.N + s;
            if constexpr (Z0 == 0) load_row<n>(a.dx + row * n, z);
            else {
#pragma unroll
                for (int i = 0; i < n; ++i) z[i] = 0.f;
            }
            load_row<m>(a.du + row * m, z + n);
        }
        float xs[n], us[m], fx[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xs[i] = xb[i] + z[i];
#pragma unroll
        for (int j = 0; j < m; ++j) us[j] = ub[j] + z[n + j];

        if constexpr (MODE == IRS_SMOOTH_FIRST_ORDER) {
            float J[n * d];
            model_jacobian<Model, float>(a.p, xs, us, fx, J);
#pragma unroll
            for (int q = 0; q < n * d; ++q) acc[q] += J[q];
        } else {
            Model::template step<float>(a.p, xs, us, fx);
            float df[n];
#pragma unroll
            for (int k = 0; k < n; ++k) df[k] = fx[k] - f0[k];
            int q = 0;
#pragma unroll
            for (int i = 0; i < NZ; ++i)
#pragma unroll
                for (int j = i; j < NZ; ++j) { acc[q] = fmaf(z[Z0 + i], z[Z0 + j], acc[q]); ++q; }
#pragma unroll
            for (int i = 0; i < NZ; ++i)
#pragma unroll
                for (int k = 0; k < n; ++k) { acc[q] = fmaf(z[Z0 + i], df[k], acc[q]); ++q; }
        }
    }
    block_reduce_store<TR::P, kWaves>(acc, red, a.partial + ((size_t)t * a.nblk + blk) * TR::P);
}

// partial (T, nblk, P) f32 -> sums (T, P) f64, fixed order over blk.
__global__ void reduce_partials_kernel(const float* __restrict__ partial, double* __restrict__ sums,
                                       int nblk, int P) {
    const int t = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float* src = partial + (size_t)t * nblk * P + p;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)src[(size_t)b * P];
    sums[(size_t)t * P + p] = s;
}

struct FinalizeArgs {
    ModelParams p;
    const double* x_trj;
    const double* u_trj;
    const double* sums;
    double* At;
    double* Bt;
    double* ct;
    int* info;
    double n_total;
    int T;
};

// One wave (64 lanes) per timestep.
template <class Model, int MODE>
__global__ __launch_bounds__(64) void smooth_finalize_kernel(FinalizeArgs a) {
    using TR = SmoothTraits<Model, MODE>;
    constexpr int n = TR::n, m = TR::m, d = TR::d, NZ = TR::NZ, Z0 = TR::Z0;
    __shared__ double G[NZ][NZ + 1];
    __shared__ double H[NZ][n];
    __shared__ double sc[NZ];
    __shared__ double AB[n][d];
    __shared__ int bad;

    const int t = blockIdx.x, lane = threadIdx.x;
    const double* S = a.sums + (size_t)t * TR::P;

    double x[n], u[m], f[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = a.x_trj[(size_t)t * n + i];
#pragma unroll
    for (int j = 0; j < m; ++j) u[j] = a.u_trj[(size_t)t * m + j];

    if constexpr (MODE == IRS_SMOOTH_FIRST_ORDER) {
        Model::template step<double>(a.p, x, u, f);
        for (int q = lane; q < n * d; q += 64) AB[q / d][q % d] = S[q] / a.n_total;
        if (lane == 0) bad = 0;
    } else {
        if constexpr (MODE == IRS_SMOOTH_ZERO_ORDER_B) {
            // A = exact Jacobian at the nominal point (quasistatic_dynamics.py:254-256)
            double J[n * d];
            model_jacobian<Model, double>(a.p, x, u, f, J);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < n; ++i)
#pragma unroll
                    for (int k = 0; k < n; ++k) AB[i][k] = J[i * d + k];
            }
        } else {
            Model::template step<double>(a.p, x, u, f);
        }
        if (lane == 0) bad = 0;
        // unpack the upper-triangular Gram and the cross term
        for (int q = lane; q < NZ * NZ; q += 64) {
            int i = q / NZ, j = q % NZ;
            int r = i < j ? i : j, c = i < j ? j : i;
            G[i][j] = S[r * NZ - r * (r - 1) / 2 + (c - r)];
        }
        for (int q = lane; q < NZ * n; q += 64) H[q / n][q % n] = S[TR::NG + q];
        __syncthreads();
        // Jacobi scaling: G' = D G D, H' = D H, D = diag(G)^-1/2
        if (lane < NZ) {
            double g = G[lane][lane];
            sc[lane] = g > 0.0 ? 1.0 / sqrt(g) : 0.0;
            if (!(g > 0.0)) bad = lane + 1;
        }
        __syncthreads();
        for (int q = lane; q < NZ * NZ; q += 64) G[q / NZ][q % NZ] *= sc[q / NZ] * sc[q % NZ];
        for (int q = lane; q < NZ * n; q += 64) H[q / n][q % n] *= sc[q / n];
        __syncthreads();
        // right-looking Cholesky, lower triangle in place
        for (int j = 0; j < NZ; ++j) {
            double djj = G[j][j];
            if (!(djj > 1e-14)) {
                if (lane == 0 && bad == 0) bad = j + 1;
                djj = 1.0;
            }
            double l = sqrt(djj);
            __syncthreads();
            if (lane == j) G[j][j] = l;
            if (lane > j && lane < NZ) G[lane][j] /= l;
            __syncthreads();
            // trailing update: element (r,c), j < c <= r < NZ
            for (int q = lane; q < NZ * NZ; q += 64) {
                int r = q / NZ, c = q % NZ;
                if (c > j && r >= c) G[r][c] -= G[r][j] * G[c][j];
            }
            __syncthreads();
        }
        // one lane per right-hand side: L y = h, L' w = y;  AB[k][Z0+i] = sc_i w_i
        if (lane < n) {
            double y[NZ];
#pragma unroll
            for (int i = 0; i < NZ; ++i) {
                double s = H[i][lane];
                for (int k = 0; k < i; ++k) s -= G[i][k] * y[k];
                y[i] = s / G[i][i];
            }
#pragma unroll
            for (int i = NZ - 1; i >= 0; --i) {
                double s = y[i];
                for (int k = i + 1; k < NZ; ++k) s -= G[k][i] * y[k];
                y[i] = s / G[i][i];
            }
#pragma unroll
            for (int i = 0; i < NZ; ++i) AB[lane][Z0 + i] = y[i] * sc[i];
        }
    }
    __syncthreads();
    for (int q = lane; q < n * n; q += 64) a.At[(size_t)t * n * n + q] = AB[q / n][q % n];
    for (int q = lane; q < n * m; q += 64) a.Bt[(size_t)t * n * m + q] = AB[q / m][n + q % m];
    if (lane < n) {
        double c = f[0];
#pragma unroll
        for (int i = 0; i < n; ++i) c = (i == lane) ? f[i] : c;
        for (int i = 0; i < n; ++i) c -= AB[lane][i] * x[i];
        for (int j = 0; j < m; ++j) c -= AB[lane][n + j] * u[j];
        a.ct[(size_t)t * n + lane] = c;
    }
    if (lane == 0) a.info[t] = bad;
}

template <class Model>
__global__ __launch_bounds__(64) void exact_linearize_kernel(ModelParams p, const double* x_trj, const double* u_trj,
                                       double* At, double* Bt, double* ct, int T) {
    constexpr int n = Model::NX, m = Model::NU, d = n + m;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double x[n], u[m], f[n], J[n * d];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = x_trj[(size_t)t * n + i];
#pragma unroll
    for (int j = 0; j < m; ++j) u[j] = u_trj[(size_t)t * m + j];
    model_jacobian<Model, double>(p, x, u, f, J);
#pragma unroll
    for (int i = 0; i < n; ++i) {
        double c = f[i];
#pragma unroll
        for (int k = 0; k < n; ++k) {
            At[((size_t)t * n + i) * n + k] = J[i * d + k];
            c -= J[i * d + k] * x[k];
        }
#pragma unroll
        for (int k = 0; k < m; ++k) {
            Bt[((size_t)t * n + i) * m + k] = J[i * d + n + k];
            c -= J[i * d + n + k] * u[k];
        }
        ct[(size_t)t * n + i] = c;
    }
}

template <int n, int m>
__global__ void rng_samples_kernel(float* dx, float* du, SmoothArgs a) {
    constexpr int d = n + m;
    const int t = blockIdx.y;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.N) return;
    const unsigned long long gidx = a.sample_offset + (unsigned long long)s;
    float z[(d + 3) / 4 * 4];
#pragma unroll
    for (int j = 0; j < (d + 3) / 4; ++j) philox_normal4(gidx, (unsigned)t, (unsigned)j, a.iter, a.seed, z + 4 * j);
    const size_t row = (size_t)t * a.N + s;
#pragma unroll
    for (int i = 0; i < n; ++i) dx[row * n + i] = z[i] * a.std[i];
#pragma unroll
    for (int j = 0; j < m; ++j) du[row * m + j] = z[n + j] * a.std[n + j];
}

void plan_grid(int T, int N, int* chunk, int* nblk) {
    // >= 1 sample per lane, and no more workgroups than ~16 per CU across the grid
    int max_blk = 4096 / (T > 0 ? T : 1);
    if (max_blk < 1) max_blk = 1;
    int nb = (N + kBlock - 1) / kBlock;
    if (nb > max_blk) nb = max_blk;
    if (nb < 1) nb = 1;
    int c = (N + nb - 1) / nb;
    c = (c + kBlock - 1) / kBlock * kBlock;
    *chunk = c;
    *nblk = (N + c - 1) / c;
    if (*nblk < 1) *nblk = 1;
}

template <class Model, int MODE>
int sums_len_t() { return SmoothTraits<Model, MODE>::P; }

template <class Model>
int sums_len_m(int mode) {
    switch (mode) {
        case IRS_SMOOTH_ZERO_ORDER_AB: return sums_len_t<Model, IRS_SMOOTH_ZERO_ORDER_AB>();
        case IRS_SMOOTH_FIRST_ORDER: return sums_len_t<Model, IRS_SMOOTH_FIRST_ORDER>();
        case IRS_SMOOTH_ZERO_ORDER_B: return sums_len_t<Model, IRS_SMOOTH_ZERO_ORDER_B>();
    }
    return -1;
}

template <class Model, bool RNG>
int launch_accum(int mode, const SmoothArgs& a, hipStream_t st) {
    dim3 grid(a.nblk, a.T), block(kBlock);
    switch (mode) {
        case IRS_SMOOTH_ZERO_ORDER_AB:
            hipLaunchKernelGGL((smooth_accum_kernel<Model, IRS_SMOOTH_ZERO_ORDER_AB, RNG>), grid, block, 0, st, a);
            break;
        case IRS_SMOOTH_FIRST_ORDER:
            hipLaunchKernelGGL((smooth_accum_kernel<Model, IRS_SMOOTH_FIRST_ORDER, RNG>), grid, block, 0, st, a);
            break;
        case IRS_SMOOTH_ZERO_ORDER_B:
            hipLaunchKernelGGL((smooth_accum_kernel<Model, IRS_SMOOTH_ZERO_ORDER_B, RNG>), grid, block, 0, st, a);
            break;
        default: return IRS_ERR_UNSUPPORTED;
    }
    return IRS_OK;
}

template <class Model>
int launch_finalize(int mode, const FinalizeArgs& a, hipStream_t st) {
    dim3 grid(a.T), block(64);
    switch (mode) {
        case IRS_SMOOTH_ZERO_ORDER_AB:
            hipLaunchKernelGGL((smooth_finalize_kernel<Model, IRS_SMOOTH_ZERO_ORDER_AB>), grid, block, 0, st, a);
            break;
        case IRS_SMOOTH_FIRST_ORDER:
            hipLaunchKernelGGL((smooth_finalize_kernel<Model, IRS_SMOOTH_FIRST_ORDER>), grid, block, 0, st, a);
            break;
        case IRS_SMOOTH_ZERO_ORDER_B:
            hipLaunchKernelGGL((smooth_finalize_kernel<Model, IRS_SMOOTH_ZERO_ORDER_B>), grid, block, 0, st, a);
            break;
        default: return IRS_ERR_UNSUPPORTED;
    }
    return IRS_OK;
}

int accumulate_common(int model, const double* params, int n_params, int mode, int T, int N,
                      const double* x_trj, const double* u_trj, SmoothArgs& a, bool rng,
                      double* sums, void* workspace, size_t workspace_bytes, void* stream) {
    IRS_CHECK_ARG(T > 0 && N > 0, "T and N must be positive");
    IRS_CHECK_ARG(x_trj && u_trj && sums && workspace, "null pointer");
    IRS_CHECK_ARG(mode >= 0 && mode <= 2, "unknown smoothing mode");
    int rc = irs_load_params(model, params, n_params, &a.p);
    if (rc != IRS_OK) return rc;
    size_t need = irs_smooth_workspace_bytes(model, mode, T, N);
    if (workspace_bytes < need) {
        irs_set_error("irs_smooth_accumulate: workspace %zu < %zu bytes", workspace_bytes, need);
        return IRS_ERR_WORKSPACE;
    }
    a.x_trj = x_trj; a.u_trj = u_trj;
    a.T = T; a.N = N;
    plan_grid(T, N, &a.chunk, &a.nblk);
    a.partial = static_cast<float*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int P = irs_sums_len(model, mode);
    IRS_DISPATCH_MODEL(model, {
        rc = rng ? launch_accum<Model, true>(mode, a, st) : launch_accum<Model, false>(mode, a, st);
    });
    if (rc != IRS_OK) return rc;
    IRS_CHECK_LAUNCH();
    dim3 g2((P + 63) / 64, T);
    hipLaunchKernelGGL(reduce_partials_kernel, g2, dim3(64), 0, st, a.partial, sums, a.nblk, P);
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

}  // namespace

extern "C" {

int irs_sums_len(int model, int mode) {
    if (mode < 0 || mode > 2) { irs_set_error("irs_sums_len: unknown mode %d", mode); return IRS_ERR_INVALID_ARG; }
    IRS_DISPATCH_MODEL(model, { return sums_len_m<Model>(mode); });
    return IRS_ERR_UNSUPPORTED;
}

size_t irs_smooth_workspace_bytes(int model, int mode, int T, int N) {
    int P = irs_sums_len(model, mode);
    if (P <= 0 || T <= 0 || N <= 0) return 0;
    int chunk, nblk;
    plan_grid(T, N, &chunk, &nblk);
    return (size_t)T * nblk * P * sizeof(float);
}

int irs_smooth_accumulate(int model, const double* params, int n_params, int mode, int T, int N,
                          const double* x_trj, const double* u_trj, const float* dx,
                          const float* du, double* sums, void* workspace,
                          size_t workspace_bytes, void* stream) {
    IRS_CHECK_ARG(du != nullptr, "du is null");
    IRS_CHECK_ARG(dx != nullptr || mode == IRS_SMOOTH_ZERO_ORDER_B, "dx is null");
    IRS_CHECK_ARG((reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (reinterpret_cast<uintptr_t>(du) & 15) == 0,
                  "dx/du must be 16-byte aligned");
    SmoothArgs a;
    memset(&a, 0, sizeof(a));
    a.dx = dx; a.du = du;
    return accumulate_common(model, params, n_params, mode, T, N, x_trj, u_trj, a, false, sums,
                             workspace, workspace_bytes, stream);
}

int irs_smooth_accumulate_rng(int model, const double* params, int n_params, int mode, int T, int N,
                              const double* x_trj, const double* u_trj, const double* std_x,
                              const double* std_u, uint64_t seed, uint32_t iter,
                              uint64_t sample_offset, double* sums, void* workspace,
                              size_t workspace_bytes, void* stream) {
    IRS_CHECK_ARG(std_u != nullptr, "std_u is null");
    IRS_CHECK_ARG(std_x != nullptr || mode == IRS_SMOOTH_ZERO_ORDER_B, "std_x is null");
    int n, m, np;
    int rc = irs_model_info(model, &n, &m, &np);
    if (rc != IRS_OK) return rc;
    SmoothArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) a.std[i] = std_x ? (float)std_x[i] : 0.f;
    for (int j = 0; j < m; ++j) a.std[n + j] = (float)std_u[j];
    a.seed = seed; a.iter = iter; a.sample_offset = sample_offset;
    return accumulate_common(model, params, n_params, mode, T, N, x_trj, u_trj, a, true, sums,
                             workspace, workspace_bytes, stream);
}

int irs_rng_samples(int n, int m, int T, int N, const double* std_x, const double* std_u,
                    uint64_t seed, uint32_t iter, uint64_t sample_offset, float* dx, float* du,
                    void* stream) {
    IRS_CHECK_ARG(T > 0 && N > 0 && dx && du && std_x && std_u, "bad argument");
    SmoothArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) a.std[i] = (float)std_x[i];
    for (int j = 0; j < m; ++j) a.std[n + j] = (float)std_u[j];
    a.seed = seed; a.iter = iter; a.sample_offset = sample_offset; a.T = T; a.N = N;
    dim3 grid((N + 255) / 256, T), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 2 && m == 1) hipLaunchKernelGGL((rng_samples_kernel<2, 1>), grid, block, 0, st, dx, du, a);
    else if (n == 12 && m == 4) hipLaunchKernelGGL((rng_samples_kernel<12, 4>), grid, block, 0, st, dx, du, a);
    else { irs_set_error("irs_rng_samples: unsupported (n,m)=(%d,%d)", n, m); return IRS_ERR_UNSUPPORTED; }
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_smooth_finalize(int model, const double* params, int n_params, int mode, int T,
                        long long N_total, const double* x_trj, const double* u_trj,
                        const double* sums, double* At, double* Bt, double* ct, int* info,
                        void* stream) {
    IRS_CHECK_ARG(T > 0 && N_total > 0, "T and N_total must be positive");
    IRS_CHECK_ARG(x_trj && u_trj && sums && At && Bt && ct && info, "null pointer");
    IRS_CHECK_ARG(mode >= 0 && mode <= 2, "unknown smoothing mode");
    FinalizeArgs a;
    int rc = irs_load_params(model, params, n_params, &a.p);
    if (rc != IRS_OK) return rc;
    a.x_trj = x_trj; a.u_trj = u_trj; a.sums = sums;
    a.At = At; a.Bt = Bt; a.ct = ct; a.info = info;
    a.n_total = (double)N_total; a.T = T;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, { rc = launch_finalize<Model>(mode, a, st); });
    if (rc != IRS_OK) return rc;
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_exact_linearize(int model, const double* params, int n_params, int T, const double* x_trj,
                        const double* u_trj, double* At, double* Bt, double* ct, void* stream) {
    IRS_CHECK_ARG(T > 0 && x_trj && u_trj && At && Bt && ct, "bad argument");
    ModelParams p;
    int rc = irs_load_params(model, params, n_params, &p);
    if (rc != IRS_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, {
        hipLaunchKernelGGL((exact_linearize_kernel<Model>), dim3((T + 63) / 64), dim3(64), 0, st, p,
                           x_trj, u_trj, At, Bt, ct, T);
    });
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

}  // extern "C"
