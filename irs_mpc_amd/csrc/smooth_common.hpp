// Shared pieces of the randomised-smoothing kernels (smooth.hip: the general sample pass; smooth_ug.hip: the
// uniform-geometry pass of the exact 8-row contact models): traits of the statistics layout, the launch
// arguments, the per-timestep solve (finalize_timestep) and the inter-workgroup hand-off (smooth_finish).
#pragma once
#include <type_traits>
#include <cstdlib>
#include "irs_common.hpp"
#include "philox.hpp"
#include "reduce.hpp"

// launch arguments of every smoothing kernel (a plain struct: shared by the translation units)
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
    int T, N, chunk, nblk, block;
    int diag;          // tuning experiments only (IRS_DIAG): 1 = skip the fused solve
    int* counters;     // (T) arrival counters, zero between calls
    float* partial;    // (T, nblk, P)
    double* fnom;      // (T, n) f64 nominal steps written by workgroup 0 when chunk0 < chunk
    int chunk0;        // samples of workgroup 0 (== chunk unless it also evaluates the nominal step)
    int wg0_rr;        // parked-sample kernels: trips of workgroup 0 dealt to all its waves; afterwards its last wave
                       // (which evaluates the nominal step) sits out (INT_MAX: plain round robin)
    double* sums;      // (T, P) out
    // finalize outputs (fused path only)
    double* At;
    double* Bt;
    double* ct;
    int* info;
    double n_total;
};

// smooth_ug.hip: the uniform-geometry sample pass (exact 8-row contact models, u-only modes)
bool irs_smooth_ug_supported(int model, int mode);
int irs_smooth_ug_nblk(int T, int N);
int irs_smooth_ug_launch(int model, int mode, const SmoothArgs& a, bool rng, bool fuse, hipStream_t st);

namespace {

constexpr int kBlock = 256;

constexpr int kCounterBytes = 4096;   // head of the workspace: arrival counters
// then T rows of f64 nominal steps (n <= 32)
constexpr size_t fnom_bytes(int T) { return (size_t)T * 32 * sizeof(double); }

// models whose step is expensive and has no Jacobian (contact QPs): in the fused launch workgroup
// 0 of every timestep takes fewer samples and evaluates the f64 nominal step the solve needs while
// the other workgroups are still sampling, instead of the last arriver doing it serially
template <class Model, int MODE>
constexpr bool nominal_in_wg0() { return !Model::HAS_JACOBIAN; }
constexpr int kNominalCost = 3;      // the f64 nominal step costs about this many f32 sample evaluations

// models with a hand-derived Jacobian in compact form (models.hpp: NJ, step_jac, expand_jac)
template <class M, class = void>
struct has_compact_jac : std::false_type {};
template <class M>
struct has_compact_jac<M, std::void_t<decltype(M::NJ)>> : std::true_type {};
template <class M>
constexpr int compact_jac_len() {
    if constexpr (has_compact_jac<M>::value) return M::NJ;
    else return 0;
}

// waves per SIMD the sample pass is compiled for (the second __launch_bounds__ argument of HIP).  A first-order pass
// with the compact hand-derived Jacobian needs ~100 registers in its loop; what would hold the FUSED kernel at two
// waves per SIMD is the one f64 nominal step of its solve (218 registers), run by one wave per timestep at the very
// end -- that code is told to fit 128 registers (it spills a little, once) so that the loop runs four waves deep and
// hides the latency of its own sample loads.
#ifndef IRS_FO_WAVES
#define IRS_FO_WAVES 1
#endif
template <class Model, int MODE>
constexpr int smooth_min_waves() {
    return (MODE == IRS_SMOOTH_FIRST_ORDER && has_compact_jac<Model>::value) ? IRS_FO_WAVES : 1;
}

template <class Model, int MODE>
struct SmoothTraits {
    static constexpr int n = Model::NX, m = Model::NU, d = n + m;
    // perturbed components that enter the least-squares design matrix
    static constexpr int NZ = (MODE == IRS_SMOOTH_ZERO_ORDER_B) ? m : d;
    // first perturbed component: n when only u is perturbed (ZERO_ORDER_B; FIRST_ORDER of a contact model)
    static constexpr int Z0 = (MODE == IRS_SMOOTH_ZERO_ORDER_B || (MODE == IRS_SMOOTH_FIRST_ORDER && !Model::HAS_JACOBIAN)) ? n : 0;
    static constexpr int NG = NZ * (NZ + 1) / 2;
    // zero-order statistics: upper Gram of z and z df'.  Models with an expensive step (contact QPs)
    // take df = f(x+dx,u+du) - xb and append sum(z): the solve then subtracts the nominal step,
    // (sum z)(f(x,u) - xb)', so that no lane of the sample pass has to evaluate f(x,u) (one whole
    // sample evaluation per lane otherwise).  Cheap analytic steps keep df = f(..) - f(x,u): for them
    // the three extra accumulators cost more than the nominal evaluation (measured, pendulum).
    static constexpr bool SUMZ = MODE != IRS_SMOOTH_FIRST_ORDER && !Model::HAS_JACOBIAN;
    static constexpr int NH = NG + NZ * n;     // offset of the sum-of-z block
    // first-order statistics: the sum of the sampled Jacobians [A | B] (n x d).  Contact models perturb
    // u only (calc_AB_first_order, quasistatic_dynamics.py:193-208) and return the decoupled pair like
    // ZERO_ORDER_B, so only the n x m block B of the active-set derivative is summed
    static constexpr bool FIRST_B = MODE == IRS_SMOOTH_FIRST_ORDER && !Model::HAS_JACOBIAN;
    static constexpr int P = FIRST_B ? n * m
                             : (MODE == IRS_SMOOTH_FIRST_ORDER) ? n * d : NG + NZ * n + (SUMZ ? NZ : 0);
    static constexpr int PP = irs_reduce_pad(P);
    // last-arriver reduction: NGRP groups of P lanes each sum a strided subset of blocks
    static constexpr int NGRP = (P >= kBlock) ? 1 : kBlock / P;
    // light = few accumulators AND a small functor: fits 1024-thread workgroups (128 VGPRs)
    static constexpr bool LIGHT = PP <= 32 && d <= 7 && Model::HAS_JACOBIAN;   // contact steps are never light
};


// Non-finite perturbations under -ffinite-math-only (these translation units are built with it): a clamp or a compare
// inside a contact step can swallow a NaN / Inf sample, and the optimiser may assume float values finite, so both the
// test and the marker go through opaque register moves.  A marked accumulator makes the solve report info != 0.
__device__ __forceinline__ bool irs_nonfinite_bits(float v) {
    unsigned bits = __float_as_uint(v);
    asm volatile("" : "+v"(bits));
    return (bits & 0x7f800000u) == 0x7f800000u;
}
__device__ __forceinline__ float irs_poison() {
    unsigned b = 0x7fc00000u;
    asm volatile("" : "+v"(b));
    return __uint_as_float(b);
}

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

// Orders this wave's LDS traffic (the LDS executes one wave's operations in issue
// order; this only stops the compiler from moving them) -- a barrier for ONE wave.
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// 1/sqrt(x) to ~1 ulp: hardware estimate + three Newton steps.  A correctly rounded f64
// sqrt followed by a divide is ~70 dependent instructions on the critical path of EVERY
// elimination step of the in-kernel solve.
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const double h = 0.5 * x * y;
        y = fma(fma(-h, y, 0.5), y, y);
    }
    return y;
}

template <class Model, int MODE>
struct FinalizeLds {
    using TR = SmoothTraits<Model, MODE>;
    double G[TR::NZ][TR::NZ + 1];
    double H[TR::NZ][TR::n];
    double sc[TR::NZ];
    double fd[TR::n];            // f(x_t,u_t) - (the f32-rounded) x_t
    double AB[TR::n][TR::d];
    int bad;
};

// Solve step for timestep t, executed by ONE wave (lane = 0..63).  S: the P f64 sums of
// the timestep (LDS or global).
template <class Model, int MODE, bool FNOM_ONLY = false>
__device__ __forceinline__ void finalize_timestep(const ModelParams& p, const double* x_trj,
                                                  const double* u_trj, const double* S, double n_total,
                                                  int t, int lane, FinalizeLds<Model, MODE>& L,
                                                  double* At, double* Bt, double* ct, int* info,
                                                  const double* fnom = nullptr) {
    using TR = SmoothTraits<Model, MODE>;
    constexpr int n = TR::n, m = TR::m, d = TR::d, NZ = TR::NZ, Z0 = TR::Z0;
    // f(x_t, u_t) in f64: evaluated here, or -- for models whose step is expensive (contact QPs) --
    // already evaluated by workgroup 0 of the fused launch, which takes fewer samples in exchange
    auto nominal = [&](const double* x, const double* u, double* f) {
        if constexpr (FNOM_ONLY) {
            // written by another wave of this launch (possibly of this workgroup): read past the vector L1
#pragma unroll
            for (int i = 0; i < n; ++i) f[i] = __hip_atomic_load(fnom + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (fnom != nullptr) {
#pragma unroll
                for (int i = 0; i < n; ++i) f[i] = fnom[i];
            } else {
                Model::template step<double>(p, x, u, f);
            }
        }
    };

    double x[n], u[m], f[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = x_trj[(size_t)t * n + i];
#pragma unroll
    for (int j = 0; j < m; ++j) u[j] = u_trj[(size_t)t * m + j];

    if (lane == 0) L.bad = 0;
    if constexpr (TR::FIRST_B) {
        // mean of the sampled B, inside the decoupled structure (irs_lqr_quasistatic.py:275-284)
        nominal(x, u, f);
        for (int q = lane; q < n * n; q += 64) {
            int i = q / n, k = q % n;
            bool act = false;
            for (int j = 0; j < m; ++j) act = act || (Model::u_into_x(j) == k);
            L.AB[i][k] = (i == k && !act) ? 1.0 : 0.0;
        }
        for (int q = lane; q < n * m; q += 64) L.AB[q / m][n + q % m] = S[q] / n_total;
    } else if constexpr (MODE == IRS_SMOOTH_FIRST_ORDER) {
        Model::template step<double>(p, x, u, f);
        for (int q = lane; q < n * d; q += 64) L.AB[q / d][q % d] = S[q] / n_total;
    } else {
        if constexpr (MODE == IRS_SMOOTH_ZERO_ORDER_B) {
            if constexpr (Model::HAS_JACOBIAN) {
                // A = exact Jacobian at the nominal point (quasistatic_dynamics.py:254-256)
                double J[n * d];
                model_jacobian<Model, double>(p, x, u, f, J);
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < n; ++i)
#pragma unroll
                        for (int k = 0; k < n; ++k) L.AB[i][k] = J[i * d + k];
                }
            } else {
                // decouple_AB (irs_lqr_quasistatic.py:275-284): A = I with the actuated
                // columns zeroed (the actuated rows of B become I after the fit, below)
                nominal(x, u, f);
                for (int q = lane; q < n * n; q += 64) {
                    int i = q / n, k = q % n;
                    bool act = false;
                    for (int j = 0; j < m; ++j) act = act || (Model::u_into_x(j) == k);
                    L.AB[i][k] = (i == k && !act) ? 1.0 : 0.0;
                }
            }
        } else {
            nominal(x, u, f);
        }
        if constexpr (NZ <= 4) {
            // tiny system: every lane solves it in registers (no LDS round trips)
            double g[NZ][NZ], h[NZ][n], scl[NZ], lo[NZ][NZ];
            int badr = 0;
#pragma unroll
            for (int i = 0; i < NZ; ++i)
#pragma unroll
                for (int j = i; j < NZ; ++j) {
                    g[i][j] = S[i * NZ - i * (i - 1) / 2 + (j - i)];
                    g[j][i] = g[i][j];
                }
            // H = sum z (f(x+dx,u+du) - f(x,u))' = sum z (f(..) - xb)' - (sum z)(f(x,u) - xb)', xb = the
            // f32-rounded nominal state the sample pass subtracted
#pragma unroll
            for (int i = 0; i < NZ; ++i)
#pragma unroll
                for (int k = 0; k < n; ++k) {
                    h[i][k] = S[TR::NG + i * n + k];
                    if constexpr (TR::SUMZ) h[i][k] -= S[TR::NH + i] * (f[k] - (double)(float)x[k]);
                }
#pragma unroll
            for (int i = 0; i < NZ; ++i) {
                bool pos = g[i][i] > 0.0;
                scl[i] = pos ? fast_rsqrt(g[i][i]) : 0.0;
                if (!pos && badr == 0) badr = i + 1;
            }
#pragma unroll
            for (int j = 0; j < NZ; ++j) {
                double djj = g[j][j] * scl[j] * scl[j];
#pragma unroll
                for (int k = 0; k < j; ++k) djj -= lo[j][k] * lo[j][k];
                if (!(djj > 1e-14)) {
                    if (badr == 0) badr = j + 1;
                    djj = 1.0;
                }
                const double il = fast_rsqrt(djj);
                lo[j][j] = il;                       // the diagonal keeps 1/l_jj
#pragma unroll
                for (int i = j + 1; i < NZ; ++i) {
                    double s = g[i][j] * scl[i] * scl[j];
#pragma unroll
                    for (int k = 0; k < j; ++k) s -= lo[i][k] * lo[j][k];
                    lo[i][j] = s * il;
                }
            }
#pragma unroll
            for (int k = 0; k < n; ++k) {
                double y[NZ];
#pragma unroll
                for (int i = 0; i < NZ; ++i) {
                    double s = h[i][k] * scl[i];
#pragma unroll
                    for (int l = 0; l < i; ++l) s -= lo[i][l] * y[l];
                    y[i] = s * lo[i][i];
                }
#pragma unroll
                for (int i = NZ - 1; i >= 0; --i) {
                    double s = y[i];
#pragma unroll
                    for (int l = i + 1; l < NZ; ++l) s -= lo[l][i] * y[l];
                    y[i] = s * lo[i][i];
                }
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < NZ; ++i) L.AB[k][Z0 + i] = y[i] * scl[i];
                }
            }
            if (lane == 0 && badr != 0) L.bad = badr;
        } else {
        // unpack the upper-triangular Gram and the cross term
        for (int q = lane; q < NZ * NZ; q += 64) {
            int i = q / NZ, j = q % NZ;
            int r = i < j ? i : j, c = i < j ? j : i;
            L.G[i][j] = S[r * NZ - r * (r - 1) / 2 + (c - r)];
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < n; ++k) L.fd[k] = f[k] - (double)(float)x[k];
        }
        wave_sync();
        for (int q = lane; q < NZ * n; q += 64) {
            double hq = S[TR::NG + q];
            if constexpr (TR::SUMZ) hq -= S[TR::NH + q / n] * L.fd[q % n];
            L.H[q / n][q % n] = hq;
        }
        wave_sync();
        // Jacobi scaling: G' = D G D, H' = D H, D = diag(G)^-1/2
        if (lane < NZ) {
            double g = L.G[lane][lane];
            L.sc[lane] = g > 0.0 ? 1.0 / sqrt(g) : 0.0;
            if (!(g > 0.0)) L.bad = lane + 1;
        }
        wave_sync();
        for (int q = lane; q < NZ * NZ; q += 64) L.G[q / NZ][q % NZ] *= L.sc[q / NZ] * L.sc[q % NZ];
        for (int q = lane; q < NZ * n; q += 64) L.H[q / n][q % n] *= L.sc[q / n];
        wave_sync();
        // Right-looking Cholesky of G' with the forward substitution folded in: the
        // right-hand sides H' ride along as extra columns (after step j row j of H' holds
        // y_j and the rows below have had L[r][j] y_j removed), then a column-oriented
        // back substitution.  Every update is spread over the 64 lanes; nothing is
        // unrolled (unrolling the solves hoists NZ^2/2 factor loads into registers and
        // would cap the whole fused sample kernel at one wave per SIMD).
        static_assert(NZ <= 32 && n <= 32, "lane split of the scaling step");
        for (int j = 0; j < NZ; ++j) {
            double djj = L.G[j][j];
            if (!(djj > 1e-14)) {
                if (lane == 0 && L.bad == 0) L.bad = j + 1;
                djj = 1.0;
            }
            const double il = fast_rsqrt(djj);
            wave_sync();
            if (lane == j) L.G[j][j] = il;                            // the diagonal keeps 1/l_jj
            if (lane > j && lane < NZ) L.G[lane][j] *= il;            // L[r][j]
            if (lane >= 32 && lane < 32 + n) L.H[j][lane - 32] *= il; // y_j
            wave_sync();
            for (int q = lane; q < NZ * NZ; q += 64) {                // (r,c), j < c <= r
                int r = q / NZ, c = q % NZ;
                if (c > j && r >= c) L.G[r][c] -= L.G[r][j] * L.G[c][j];
            }
            for (int q = lane; q < NZ * n; q += 64) {                 // rows below j of H'
                int r = q / n, k = q % n;
                if (r > j) L.H[r][k] -= L.G[r][j] * L.H[j][k];
            }
            wave_sync();
        }
        // L' w = y, column oriented
        for (int i = NZ - 1; i >= 0; --i) {
            const double il = L.G[i][i];
            wave_sync();
            if (lane < n) L.H[i][lane] *= il;                         // w_i
            wave_sync();
            for (int q = lane; q < i * n; q += 64) {
                int r = q / n, k = q % n;
                L.H[r][k] -= L.G[i][r] * L.H[i][k];
            }
        }
        wave_sync();
        for (int q = lane; q < NZ * n; q += 64) L.AB[q % n][Z0 + q / n] = L.H[q / n][q % n] * L.sc[q / n];
        }  // NZ > 4
    }
    wave_sync();
    if constexpr ((MODE == IRS_SMOOTH_ZERO_ORDER_B && !Model::HAS_JACOBIAN) || TR::FIRST_B) {
        for (int q = lane; q < m * m; q += 64) L.AB[Model::u_into_x(q / m)][n + q % m] = (q / m == q % m) ? 1.0 : 0.0;
        wave_sync();
    }
    for (int q = lane; q < n * n; q += 64) At[(size_t)t * n * n + q] = L.AB[q / n][q % n];
    for (int q = lane; q < n * m; q += 64) Bt[(size_t)t * n * m + q] = L.AB[q / m][n + q % m];
    if (lane < n) {
        double c = f[0];
#pragma unroll
        for (int i = 0; i < n; ++i) c = (i == lane) ? f[i] : c;
#pragma unroll
        for (int i = 0; i < n; ++i) c -= L.AB[lane][i] * x[i];
#pragma unroll
        for (int j = 0; j < m; ++j) c -= L.AB[lane][n + j] * u[j];
        ct[(size_t)t * n + lane] = c;
    }
    // Non-finite statistics (an f32 sample that diverged): tested on the BIT PATTERN -- this translation
    // unit is built with -ffinite-math-only, under which `!(d > eps)` may be lowered to an ordered compare
    // that a NaN passes.  info = P + 1 marks it (the pivot codes are 1..NZ).
    {
        bool nonfinite = false;
        for (int q = lane; q < TR::P; q += 64) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(S[q]);
            nonfinite = nonfinite || ((bits & 0x7ff0000000000000ull) == 0x7ff0000000000000ull);
        }
        if (__any(nonfinite) && lane == 0) L.bad = TR::P + 1;
    }
    wave_sync();
    if (lane == 0) info[t] = L.bad;
}

template <class Model, class = void>
struct contact_rows_of { static constexpr int value = 0; };
template <class Model>
struct contact_rows_of<Model, std::void_t<decltype(Model::NC)>> { static constexpr int value = Model::NC; };
template <class Model>
constexpr int contact_rows() { return contact_rows_of<Model>::value; }


// What every smoothing kernel does once its workgroup's per-wave partial statistics sit in `red` (NW rows of
// TR::PP floats): sum the rows, publish the workgroup's partial (write-through 16-byte stores), take a ticket;
// the last arriver of the timestep adds the nblk partials in a FIXED order in f64 -> sums[t] and (FUSE) solves.
// `fnom_t`: the f64 nominal step of this timestep if a wave of the launch evaluated it (else the solve does).
// FNOM_ONLY: the solve never evaluates the model's step itself (kernels that must stay register-lean).
template <class Model, int MODE, bool FUSE, int BLOCK, bool FNOM_ONLY = false>
__device__ __forceinline__ void smooth_finish(const SmoothArgs& a, float* red, double* red64, double* tot,
                                              FinalizeLds<Model, MODE>& fin, int& s_ticket, int t, int blk, int tid,
                                              const double* fnom_t) {
    using TR = SmoothTraits<Model, MODE>;
    constexpr int n = TR::n, P = TR::P;
    constexpr int NW = BLOCK / 64;
    constexpr int P4 = (P + 3) / 4 * 4;
    (void)n;
    __syncthreads();

    const int nblk = a.nblk;
    if (nblk == 1) {
        // the only workgroup of this timestep: totals straight from LDS
        for (int q = tid; q < P; q += BLOCK) {
            float s = red[q];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w * TR::PP + q];
            tot[q] = (double)s;
        }
    } else if constexpr (BLOCK <= 512 || FNOM_ONLY) {
        // publish this workgroup's partial sums: 16-byte write-through (sc1) stores, one
        // row of P4 floats per workgroup (few wide fabric writes instead of P narrow ones)
        for (int q = tid; q < P; q += BLOCK) {
            float s = red[q];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[w * TR::PP + q];
            red[q] = s;                       // column q is touched by this lane only
        }
        __syncthreads();
        {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const unsigned bytes = (unsigned)((size_t)a.T * nblk * P4 * sizeof(float));
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.partial, 0, bytes, 0x00020000);
            const unsigned row = (unsigned)(((size_t)t * nblk + blk) * P4 * sizeof(float));
            for (int q4 = tid; q4 < P4 / 4; q4 += BLOCK) {
                u32x4 v;
                v.x = __float_as_uint(red[4 * q4]);
                v.y = __float_as_uint(red[4 * q4 + 1]);
                v.z = __float_as_uint(red[4 * q4 + 2]);
                v.w = __float_as_uint(red[4 * q4 + 3]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, row + 16u * q4, 0, 16 /* sc1 */);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            s_ticket = __hip_atomic_fetch_add(a.counters + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != nblk - 1) return;       // uniform over the workgroup
        // ---- last arriver of timestep t ------------------------------------------
        if (tid == 0) {
            __hip_atomic_store(a.counters + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* src = a.partial + (size_t)t * nblk * P4;
        if constexpr (P >= kBlock) {
            // plain loads are valid behind the acquire above; unrolled so that the
            // (independent) loads are in flight together, summed in a fixed order
            for (int q = tid; q < P; q += BLOCK) {
                double s = 0.0;
#pragma unroll 8
                for (int b = 0; b < nblk; ++b) s += (double)src[(size_t)b * P4 + q];
                tot[q] = s;
            }
        } else {
            const int g = tid / P, q = tid % P;
            if (g < TR::NGRP) {
                double s = 0.0;
#pragma unroll 4
                for (int b = g; b < nblk; b += TR::NGRP) s += (double)src[(size_t)b * P4 + q];
                red64[g * P + q] = s;
            }
            __syncthreads();
            if (tid < P) {
                double s = red64[tid];
                for (int gg = 1; gg < TR::NGRP; ++gg) s += red64[gg * P + tid];
                tot[tid] = s;
            }
        }
    }
    __syncthreads();
    for (int q = tid; q < P; q += BLOCK) a.sums[(size_t)t * P + q] = tot[q];
    if constexpr (FUSE) {
        if (tid < 64 && a.diag != 1)
            finalize_timestep<Model, MODE, FNOM_ONLY>(a.p, a.x_trj, a.u_trj, tot, a.n_total, t, tid, fin, a.At, a.Bt,
                                           a.ct, a.info,
                                           fnom_t);
    }
}

}  // namespace
