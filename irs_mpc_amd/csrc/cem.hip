// Cross-entropy-method baseline: CrossEntropyMethod.local_descent (irs_lqr/cem.py:151-184).
//   1. roll out every candidate control sequence u_cand[b] (B of them) for T steps on the
//      true dynamics and evaluate its cost            (cem.py:163-168; the ONLY place the
//      reference batches multi-step rollouts)         -> cem_rollout_kernel, one lane per b
//   2. keep the n_elite cheapest (np.argpartition, :173) -> cem_select_kernel: radix select
//      on order-preserving 64-bit keys + ordered compaction (deterministic)
//   3. refit mean / std over the elites (:178-180)      -> cem_refit_kernel
//   4. roll out the mean (:182)                          -> rollout_kernel (tvlqr.hip)
// All arithmetic in f64: rollout costs only rank the candidates, but ties broken by f32
// noise would change the elite set and hence the refit.
#include "boxqp.hpp"      // has_u_into_x

namespace {

template <class Model>
__global__ __launch_bounds__(256) void cem_rollout_kernel(ModelParams p, int T, int B,
                                                          const double* __restrict__ u_cand,
                                                          const double* __restrict__ x0,
                                                          const double* __restrict__ Q,
                                                          const double* __restrict__ R,
                                                          const double* __restrict__ xd_trj,
                                                          double* __restrict__ costs) {
    constexpr int n = Model::NX, m = Model::NU;
    __shared__ double Qs[n * n];
    __shared__ double Rs[m * m];
    for (int q = threadIdx.x; q < n * n; q += blockDim.x) Qs[q] = Q[q];
    for (int q = threadIdx.x; q < m * m; q += blockDim.x) Rs[q] = R[q];
    __syncthreads();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x[n], u[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = x0[i];
    const double* ub = u_cand + (size_t)b * T * m;
    double cost = 0.0;
    unsigned warm_set = ~0u;          // active set of the previous contact step (exact step QPs only)
    for (int t = 0; t <= T; ++t) {
        // (x_t - xd_t)' Q (x_t - xd_t); the terminal term also uses Q (cem.py:138-139)
        const double* xd = xd_trj + (size_t)t * n;      // uniform address: scalar loads
        double e[n];
#pragma unroll
        for (int i = 0; i < n; ++i) e[i] = x[i] - xd[i];
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) r += Qs[i * n + j] * e[j];
            cost += e[i] * r;
        }
        if (t == T) break;
#pragma unroll
        for (int j = 0; j < m; ++j) u[j] = ub[(size_t)t * m + j];
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) r += Rs[i * m + j] * u[j];
            cost += u[i] * r;
        }
        irs_step_along<Model>(p, x, u, xn, &warm_set);
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xn[i];
    }
    costs[b] = cost;
}

// CrossEntropyMethodQuasistatic.local_descent steps 1-2 (irs_lqr/cem_quasistatic.py:186-200): the
// candidate cost is IrsLqrQuasistatic's eval_cost (:124-165) -- state error with Q, TERMINAL Qd,
// input cost on du_t = u_t - u_{t-1} with du_0 = u_0 - x_0[indices_u_into_x].
template <class Model>
__global__ __launch_bounds__(64) void cem_rollout_quasistatic_kernel(ModelParams p, int T, int B,
                                                                     const double* __restrict__ u_cand,
                                                                     const double* __restrict__ x0,
                                                                     const double* __restrict__ Q,
                                                                     const double* __restrict__ Qd,
                                                                     const double* __restrict__ R,
                                                                     const double* __restrict__ xd_trj,
                                                                     double* __restrict__ costs) {
    constexpr int n = Model::NX, m = Model::NU;
    __shared__ double Qs[n * n];
    __shared__ double Qds[n * n];
    __shared__ double Rs[m * m];
    for (int q = threadIdx.x; q < n * n; q += blockDim.x) { Qs[q] = Q[q]; Qds[q] = Qd[q]; }
    for (int q = threadIdx.x; q < m * m; q += blockDim.x) Rs[q] = R[q];
    __syncthreads();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x[n], u[m], up[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = x0[i];
#pragma unroll
    for (int j = 0; j < m; ++j) up[j] = x0[Model::u_into_x(j)];
    const double* ub = u_cand + (size_t)b * T * m;
    double cost = 0.0;
    unsigned warm_set = ~0u;          // active set of the previous contact step (exact step QPs only)
    for (int t = 0; t <= T; ++t) {
        const double* xd = xd_trj + (size_t)t * n;
        const double* W = t == T ? Qds : Qs;
        double e[n];
#pragma unroll
        for (int i = 0; i < n; ++i) e[i] = x[i] - xd[i];
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) r += W[i * n + j] * e[j];
            cost += e[i] * r;
        }
        if (t == T) break;
        double dv[m];
#pragma unroll
        for (int j = 0; j < m; ++j) { u[j] = ub[(size_t)t * m + j]; dv[j] = u[j] - up[j]; up[j] = u[j]; }
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) r += Rs[i * m + j] * dv[j];
            cost += dv[i] * r;
        }
        irs_step_along<Model>(p, x, u, xn, &warm_set);
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = xn[i];
    }
    costs[b] = cost;
}

// Order-preserving map f64 -> u64 (NaN sorts last: a diverged rollout is never elite).
__device__ __forceinline__ unsigned long long cost_key(double c) {
    if (c != c) return ~0ull;
    unsigned long long b = (unsigned long long)__double_as_longlong(c);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

constexpr int kSelBlock = 1024;

// Single workgroup.  Finds the n_elite smallest costs; writes their indices in increasing
// index order (ties at the threshold: lowest indices first) -> elite_idx[0..n_elite).
__global__ __launch_bounds__(kSelBlock) void cem_select_kernel(const double* __restrict__ costs, int B,
                                                               int n_elite, int* __restrict__ elite_idx) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_k, s_binc;
    __shared__ int scan[kSelBlock];
    __shared__ int s_less_total;
    const int tid = threadIdx.x;
    if (tid == 0) { s_prefix = 0ull; s_k = n_elite; }
    __syncthreads();
    // MSB-first radix select of the n_elite-th smallest key
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        const unsigned long long mask = pass == 0 ? 0ull : (~0ull << (shift + 8));
        for (int i = tid; i < B; i += kSelBlock) {
            unsigned long long key = cost_key(costs[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            // the bin holding the k-th smallest candidate: inclusive prefix sums of the 256 counts, four per lane
            // (one wave; the serial scan by one lane was 8 x 256 dependent LDS reads of the kernel's 74 us)
            const int k = s_k;
            int c[4], run = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c[j] = (int)hist[4 * tid + j]; run += c[j]; }
            int incl = run;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                if (tid >= off) incl += v;
            }
            int before = incl - run;                   // candidates in the bins of lower lanes
            const bool mine = before < k && k <= incl; // exactly one lane
            if (mine) {
                int bin = 4 * tid, kk = k - before;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (kk > c[j] && j < 3) { kk -= c[j]; ++bin; } else break;
                }
                s_k = kk;                              // rank inside the chosen bin
                s_prefix = prefix | ((unsigned long long)bin << shift);
                s_binc = (int)hist[bin];
            }
        }
        __syncthreads();
        // every key of the chosen bin is elite: no need to refine further -- the largest key of the bin is a valid
        // threshold (keys equal to it, if any, all belong)
        const bool whole_bin = s_k == s_binc && pass < 7;
        __syncthreads();                               // everyone has read the pair before lane 0 rewrites it
        if (whole_bin) {
            if (tid == 0) {
                s_prefix |= (1ull << shift) - 1ull;
                s_k = 0x7fffffff;
            }
            __syncthreads();
            break;
        }
    }
    const unsigned long long thr = s_prefix;           // key of the n_elite-th smallest cost (or the top of its bin)
    const int need_equal = s_k;                        // how many keys == thr belong to the elite
    // ordered compaction: thread owns a contiguous chunk of indices
    const int chunk = (B + kSelBlock - 1) / kSelBlock;
    const int lo = tid * chunk, hi = min(B, lo + chunk);
    int n_less = 0, n_eq = 0;
    for (int i = lo; i < hi; ++i) {
        unsigned long long key = cost_key(costs[i]);
        n_less += key < thr;
        n_eq += key == thr;
    }
    // exclusive scans of n_less and n_eq over the threads (two passes through one buffer)
    auto block_exclusive_scan = [&](int v, int* total) {
        scan[tid] = v;
        __syncthreads();
        for (int off = 1; off < kSelBlock; off <<= 1) {
            int add = tid >= off ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        int incl = scan[tid];
        if (total != nullptr && tid == kSelBlock - 1) *total = incl;
        __syncthreads();
        return incl - v;
    };
    const int less_before = block_exclusive_scan(n_less, &s_less_total);
    const int eq_before = block_exclusive_scan(n_eq, nullptr);
    const int less_total = s_less_total;               // == n_elite - need_equal
    int wl = less_before, we = eq_before;
    for (int i = lo; i < hi; ++i) {
        unsigned long long key = cost_key(costs[i]);
        if (key < thr) {
            elite_idx[wl++] = i;                        // provisional slot; merged below
        } else if (key == thr) {
            if (we < need_equal) elite_idx[less_total + we] = i;
            ++we;
        }
    }
}

// u_new = mean over elites, std_new = population std over elites (np.mean / np.std, axis 0: two passes).
// One workgroup per 64 consecutive outputs q: lane = q (an elite's row is contiguous in q: coalesced), the 16 waves
// split the elites and meet in LDS, partial sums added in wave order (deterministic).  (One lane per q looping over
// all elites alone -- 2 x n_elite dependent loads -- took 118 us for 312 elites.)
constexpr int kRefitWaves = 16;
__global__ __launch_bounds__(64 * kRefitWaves) void cem_refit_kernel(const double* __restrict__ u_cand,
                                                                     const int* __restrict__ elite_idx, int n_elite,
                                                                     int Tm, double* __restrict__ u_new,
                                                                     double* __restrict__ std_new) {
    __shared__ double part[kRefitWaves][64];
    __shared__ double mean_s[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const bool on = q < Tm;
    double s = 0.0;
    for (int e = wave; e < n_elite; e += kRefitWaves)
        s += on ? u_cand[(size_t)elite_idx[e] * Tm + q] : 0.0;
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kRefitWaves; ++w) t += part[w][lane];
        mean_s[lane] = t / n_elite;
    }
    __syncthreads();
    const double mean = mean_s[lane];
    double v = 0.0;
    for (int e = wave; e < n_elite; e += kRefitWaves) {
        const double d = on ? u_cand[(size_t)elite_idx[e] * Tm + q] - mean : 0.0;
        v += d * d;
    }
    __syncthreads();
    part[wave][lane] = v;
    __syncthreads();
    if (wave == 0 && on) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kRefitWaves; ++w) t += part[w][lane];
        u_new[q] = mean;
        std_new[q] = sqrt(t / n_elite);
    }
}

}  // namespace

extern "C" {

int irs_cem_rollout_costs(int model, const double* params, int n_params, int T, int B,
                          const double* u_cand, const double* x0, const double* Q, const double* R,
                          const double* xd_trj, double* costs, void* stream) {
    IRS_CHECK_ARG(T > 0 && B > 0 && u_cand && x0 && Q && R && xd_trj && costs, "bad argument");
    ModelParams p;
    int rc = irs_load_params(model, params, n_params, &p);
    if (rc != IRS_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, {
        hipLaunchKernelGGL((cem_rollout_kernel<Model>), dim3((B + 255) / 256), dim3(256), 0, st, p, T, B,
                           u_cand, x0, Q, R, xd_trj, costs);
    });
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_cem_rollout_costs_quasistatic(int model, const double* params, int n_params, int T, int B,
                                      const double* u_cand, const double* x0, const double* Q,
                                      const double* Qd, const double* R, const double* xd_trj, double* costs,
                                      void* stream) {
    IRS_CHECK_ARG(T > 0 && B > 0 && u_cand && x0 && Q && Qd && R && xd_trj && costs, "bad argument");
    ModelParams p;
    int rc = irs_load_params(model, params, n_params, &p);
    if (rc != IRS_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = IRS_ERR_UNSUPPORTED;
    IRS_DISPATCH_MODEL(model, {
        if constexpr (has_u_into_x<Model>::value) {
            // one wave per workgroup: the contact step holds hundreds of f64 registers per lane
            hipLaunchKernelGGL((cem_rollout_quasistatic_kernel<Model>), dim3((B + 63) / 64), dim3(64), 0, st, p,
                               T, B, u_cand, x0, Q, Qd, R, xd_trj, costs);
            rc = IRS_OK;
        } else {
            irs_set_error("irs_cem_rollout_costs_quasistatic: model %d is not position controlled", model);
        }
    });
    if (rc != IRS_OK) return rc;
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_cem_refit(int T, int m, int B, int n_elite, const double* u_cand, const double* costs,
                  int* elite_idx, double* u_new, double* std_new, void* stream) {
    IRS_CHECK_ARG(T > 0 && m > 0 && B > 0 && n_elite > 0 && n_elite <= B, "need 0 < n_elite <= B");
    IRS_CHECK_ARG(u_cand && costs && elite_idx && u_new && std_new, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(cem_select_kernel, dim3(1), dim3(kSelBlock), 0, st, costs, B, n_elite, elite_idx);
    IRS_CHECK_LAUNCH();
    const int Tm = T * m;
    hipLaunchKernelGGL(cem_refit_kernel, dim3((Tm + 63) / 64), dim3(64 * kRefitWaves), 0, st, u_cand, elite_idx, n_elite,
                       Tm, u_new, std_new);
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

}  // extern "C"
