// IrsLqr.iterate (irs_lqr/irs_lqr.py:188-218) as ONE call at the boundary: k+1 descents -- linearise (randomised
// smoothing on device-drawn samples, or exactly), Riccati + closed-loop rollout + cost, and, when box bounds are
// given, the test "does every tail's unconstrained plan stay inside the box" followed by the bounded descent
// behind a device-side flag -- enqueued back to back on the caller's stream.  Nothing returns to the host between
// phases or between iterations; descent i linearises around the trajectory descent i-1 wrote into the history
// buffers, which are read back once, at the end.
//
// The reference's loop (Python, one QP per timestep per iteration) is what this replaces; the host twin
// irs_mpc_amd/irs_lqr.py routes IrsLqr*.iterate here whenever the sampling object can be drawn on the device
// (GaussianSmoothing) or the linearisation is exact.
#include "irs_common.hpp"

namespace {

// Tail t's unconstrained plan = the policy (K_s, k_s), s >= t, rolled out on the LINEAR model from the realised
// state x_t (Bellman).  If every tail's plan respects the box, the Riccati descent is the solution of the bounded
// QPs too (tv_lqr.py:112-123); otherwise *flag = 1 and the bounded descent must run.  One thread per tail; all
// tails advance together (thread t is idle until s = t), so the coefficient rows of step s are read by every
// active thread at once.  (irs_mpc_amd/irs_lqr.py:_tail_plans_within_bounds is the host statement of the same.)
__global__ void plan_check_kernel(int n, int m, int T, const double* __restrict__ At, const double* __restrict__ Bt,
                                  const double* __restrict__ ct, const double* __restrict__ K,
                                  const double* __restrict__ k, const double* __restrict__ x_new,
                                  const double* __restrict__ xlo, const double* __restrict__ xhi,
                                  const double* __restrict__ ulo, const double* __restrict__ uhi, int* flag,
                                  const int* descent_info, const int* smooth_info, int box_unsupported, int* row) {
    constexpr int NMAX = 32, MMAX = 16;
    __shared__ int viol;
    if (threadIdx.x == 0) viol = 0;
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += blockDim.x) {
        const int t = t0 + threadIdx.x;
        double x[NMAX], u[MMAX], xn[NMAX];
        bool bad = false;
        for (int s = t0; s < T; ++s) {
            if (t < T && s == t)
                for (int i = 0; i < n; ++i) x[i] = x_new[(size_t)t * n + i];
            if (t < T && s >= t && !bad) {
                const double* Ks = K + (size_t)s * m * n;
                for (int j = 0; j < m; ++j) {
                    double a = k[(size_t)s * m + j];
                    for (int i = 0; i < n; ++i) a += Ks[j * n + i] * x[i];
                    u[j] = a;
                    bad = bad || a < ulo[j] || a > uhi[j];
                }
                const double* As = At + (size_t)s * n * n;
                const double* Bs = Bt + (size_t)s * n * m;
                for (int i = 0; i < n; ++i) {
                    double a = ct[(size_t)s * n + i];
                    for (int l = 0; l < n; ++l) a += As[i * n + l] * x[l];
                    for (int j = 0; j < m; ++j) a += Bs[i * m + j] * u[j];
                    xn[i] = a;
                    bad = bad || a < xlo[i] || a > xhi[i];
                }
                for (int i = 0; i < n; ++i) x[i] = xn[i];
            }
        }
        if (bad) viol = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) *flag = viol;
    // (fused iterate: the head of this descent's info row, so that a bounded iteration needs no separate info launch
    // unless the bounded descent can run)
    if (row != nullptr) {
        int bad = 0;
        if (smooth_info != nullptr)
            for (int t = threadIdx.x; t < T; t += blockDim.x) bad += smooth_info[t] != 0 ? 1 : 0;
        __shared__ int nbad;
        if (threadIdx.x == 0) nbad = 0;
        __syncthreads();
        if (bad) atomicAdd(&nbad, bad);
        __syncthreads();
        if (threadIdx.x == 0) {
            row[0] = descent_info[0];
            row[1] = nbad;
            row[2] = viol;
            row[3] = row[4] = row[5] = 0;
            row[6] = viol && box_unsupported ? 1 : 0;
            row[7] = 0;
        }
    }
}

// lane N of every 16-lane row to all lanes of that row (DPP row_newbcast, as in ctrlbox_mfma.hip)
template <int N>
__device__ __forceinline__ double pc_row_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + N, 0xf, 0xf, true);     // bound_ctrl: no destination initialisation
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// one unrolled step of the walk: x_l to its 16-lane group by DPP, two partial sums per output (half the FMA chain)
template <int NN, int l = 0>
struct PcWalk {
    static __device__ __forceinline__ void run(const double* a_r, const double* k_r, double x, double* ax, double* au) {
        const double xl = pc_row_bcast<l>(x);
        ax[l & 1] = fma(a_r[l], xl, ax[l & 1]);
        au[l & 1] = fma(k_r[l], xl, au[l & 1]);
        if constexpr (l + 1 < NN) PcWalk<NN, l + 1>::run(a_r, k_r, x, ax, au);
    }
};

// The same test, fast: the serial kernel above walks every tail with ONE thread that fetches (A, B, K) from L2 step
// by step -- 1.07 ms for the quadrotor (T = 50, n = 12), 50 us for the pendulum, per iteration, with the bounds the
// reference's scripts pass.  Here one 1024-thread workgroup (a) forms the closed loop of every step ONCE,
//     x+ = Acl_s x + bcl_s,  Acl_s = A_s + B_s K_s,  bcl_s = B_s k_s + c_s,
// into LDS (transposed: lanes read consecutive words), then (b) walks all tails at once, 16 lanes per tail (lane i
// owns x_i and row i; lanes < m also the control row): x_l reaches its group by one DPP move per word (a shuffle
// through LDS is ~200 cycles of latency: twelve in a row per step made this 55 us), the state size is a template
// parameter (every loop unrolled, no predication).
template <int NN>
__global__ __launch_bounds__(1024) void plan_check16_kernel(int m, int T, const double* __restrict__ At,
                                                            const double* __restrict__ Bt, const double* __restrict__ ct,
                                                            const double* __restrict__ K, const double* __restrict__ k,
                                                            const double* __restrict__ x_new,
                                                            const double* __restrict__ xlo, const double* __restrict__ xhi,
                                                            const double* __restrict__ ulo, const double* __restrict__ uhi,
                                                            int* flag, const int* descent_info, const int* smooth_info,
                                                            int box_unsupported, int* row) {
    constexpr int n = NN, L = 16;
    extern __shared__ double psh[];
    double* AclT = psh;                          // [s][l][i]
    double* bcl = AclT + (size_t)T * n * n;      // [s][i]
    double* KsT = bcl + (size_t)T * n;           // [s][l][j]
    double* ks = KsT + (size_t)T * n * m;        // [s][j]
    __shared__ int viol, nbad;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) { viol = 0; nbad = 0; }
    for (int idx = tid; idx < T * n * n; idx += nt) {
        const int s = idx / (n * n), r = idx - s * n * n, l = r / n, i = r - l * n;
        double a = At[((size_t)s * n + i) * n + l];
        for (int j = 0; j < m; ++j) a = fma(Bt[((size_t)s * n + i) * m + j], K[((size_t)s * m + j) * n + l], a);
        AclT[idx] = a;
    }
    for (int idx = tid; idx < T * n; idx += nt) {
        const int s = idx / n, i = idx - s * n;
        double b = ct[idx];
        for (int j = 0; j < m; ++j) b = fma(Bt[((size_t)s * n + i) * m + j], k[(size_t)s * m + j], b);
        bcl[idx] = b;
    }
    for (int idx = tid; idx < T * n * m; idx += nt) {
        const int s = idx / (n * m), r = idx - s * n * m, l = r / m, j = r - l * m;
        KsT[idx] = K[((size_t)s * m + j) * n + l];
    }
    for (int idx = tid; idx < T * m; idx += nt) ks[idx] = k[idx];
    __syncthreads();
    const int G = nt / L, g = tid / L, i = tid % L;
    const int ii = i < n ? i : 0, jj = i < m ? i : 0;       // lanes without a row recompute row 0 and discard it
    const double lo_x = xlo[ii], hi_x = xhi[ii], lo_u = ulo[jj], hi_u = uhi[jj];
    bool bad = false;
    // every lane of a wave runs the same trip counts: a group without a tail walks the last one and discards
    for (int tb = 0; tb < T; tb += G) {
        const int t = tb + g;
        const bool live = t < T;
        const int tt = live ? t : T - 1;
        // the four groups of a wave start at consecutive tails: walk from the wave's earliest
        int s = tb + (tid / 64) * (64 / L);
        s = s < T ? s : T - 1;
        double x = i < n ? x_new[(size_t)tt * n + i] : 0.0;
        double a_r[NN], k_r[NN], b_r, kk_r;
        auto fetch = [&](int s_) {
            const double* Ar = AclT + (size_t)s_ * n * n + ii;
            const double* Kr = KsT + (size_t)s_ * n * m + jj;
#pragma unroll
            for (int l = 0; l < NN; ++l) { a_r[l] = Ar[l * n]; k_r[l] = Kr[l * m]; }
            b_r = bcl[s_ * n + ii];
            kk_r = ks[s_ * m + jj];
        };
        for (; s < T; ++s) {
            // (no prefetch of the next step's rows: with 24 + 24 registers of operands in flight twice the kernel spills
            // under the 128-register limit of a 1024-thread workgroup; three to four waves per SIMD hide the LDS latency)
            fetch(s);
            double ax[2] = {b_r, 0.0}, au[2] = {kk_r, 0.0};
            PcWalk<NN>::run(a_r, k_r, x, ax, au);
            const double xs = ax[0] + ax[1], us = au[0] + au[1];
            if (s >= tt) {
                bad = bad || (live && i < m && (us < lo_u || us > hi_u)) || (live && i < n && (xs < lo_x || xs > hi_x));
                x = i < n ? xs : 0.0;
            }
        }
    }
    if (bad) viol = 1;
    if (smooth_info != nullptr) {
        int c = 0;
        for (int t = tid; t < T; t += nt) c += smooth_info[t] != 0 ? 1 : 0;
        if (c) atomicAdd(&nbad, c);
    }
    __syncthreads();
    if (tid == 0) {
        *flag = viol;
        if (row != nullptr) {
            row[0] = descent_info[0];
            row[1] = nbad;
            row[2] = viol;
            row[3] = row[4] = row[5] = 0;
            row[6] = viol && box_unsupported ? 1 : 0;
            row[7] = 0;
        }
    }
}

template <int NN>
static int plan_check16_launch(int m, int T, size_t bytes, const double* At, const double* Bt, const double* ct,
                               const double* K, const double* k, const double* x_new, const double* xlo,
                               const double* xhi, const double* ulo, const double* uhi, int* flag,
                               const int* descent_info, const int* smooth_info, int box_unsupported, int* row,
                               hipStream_t st) {
    static bool attr = false;
    auto kern = plan_check16_kernel<NN>;
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) {
            irs_set_error("plan check: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return IRS_ERR_HIP;
        }
        attr = true;
    }
    // as many 64-lane waves as there are groups of four tails, at most 16
    const int waves = (T + 3) / 4 < 16 ? (T + 3) / 4 : 16;
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), bytes, st, m, T, At, Bt, ct, K, k, x_new, xlo, xhi, ulo, uhi, flag,
                       descent_info, smooth_info, box_unsupported, row);
    return IRS_OK;
}

// launches the fast kernel when the closed loops fit LDS and n, m <= 16 (else the serial one)
static int plan_check_launch(int n, int m, int T, const double* At, const double* Bt, const double* ct, const double* K,
                             const double* k, const double* x_new, const double* xlo, const double* xhi,
                             const double* ulo, const double* uhi, int* flag, const int* descent_info,
                             const int* smooth_info, int box_unsupported, int* row, hipStream_t st) {
    const size_t bytes = (size_t)T * ((size_t)n * n + n + (size_t)n * m + m) * sizeof(double);
    if (n >= 1 && n <= 16 && m <= 16 && bytes <= (size_t)(150 * 1024)) {
#define PC_CASE(NN_)                                                                                                    \
    case NN_:                                                                                                           \
        return plan_check16_launch<NN_>(m, T, bytes, At, Bt, ct, K, k, x_new, xlo, xhi, ulo, uhi, flag, descent_info,   \
                                        smooth_info, box_unsupported, row, st);
        switch (n) {
            PC_CASE(1) PC_CASE(2) PC_CASE(3) PC_CASE(4) PC_CASE(5) PC_CASE(6) PC_CASE(7) PC_CASE(8)
            PC_CASE(9) PC_CASE(10) PC_CASE(11) PC_CASE(12) PC_CASE(13) PC_CASE(14) PC_CASE(15) PC_CASE(16)
        }
#undef PC_CASE
    }
    const int block = T < 256 ? ((T + 63) / 64 * 64) : 256;
    hipLaunchKernelGGL(plan_check_kernel, dim3(1), dim3(block), 0, st, n, m, T, At, Bt, ct, K, k, x_new, xlo, xhi, ulo, uhi,
                       flag, descent_info, smooth_info, box_unsupported, row);
    return IRS_OK;
}

// row of the iteration's info history: [0] Riccati info, [1] timesteps whose smoothing solve failed, [2] box needed,
// [3..5] the bounded descent's info (valid when [2] != 0), [6] box needed but the horizon does not fit its kernel.
// Written by plan_check_kernel (bounds given) or by the descent's own launch (no bounds: tvlqr.hip, descent_kernel).
struct PhaseTimer {
    bool on;
    hipStream_t st;
    hipEvent_t ev[4];
    double ms[3];
    explicit PhaseTimer(bool enable, hipStream_t s) : on(enable), st(s) {
        ms[0] = ms[1] = ms[2] = 0.0;
        if (on)
            for (auto& e : ev) (void)hipEventCreate(&e);
    }
    void mark(int i) {
        if (on) (void)hipEventRecord(ev[i], st);
    }
    // (timing mode synchronises after every iteration: it measures phases, not the pipeline)
    void collect() {
        if (!on) return;
        (void)hipEventSynchronize(ev[3]);
        for (int i = 0; i < 3; ++i) {
            float v = 0.f;
            (void)hipEventElapsedTime(&v, ev[i], ev[i + 1]);
            ms[i] += v;
        }
    }
    ~PhaseTimer() {
        if (on)
            for (auto& e : ev) (void)hipEventDestroy(e);
    }
};

}  // namespace

extern "C" {

int irs_tvlqr_plan_within_bounds(int n, int m, int T, const double* At, const double* Bt, const double* ct,
                                 const double* K, const double* k, const double* x_new, const double* xlo,
                                 const double* xhi, const double* ulo, const double* uhi, int* flag, void* stream) {
    IRS_CHECK_ARG(n > 0 && n <= 32 && m > 0 && m <= 16 && T > 0, "sizes out of range (n <= 32, m <= 16)");
    IRS_CHECK_ARG(At && Bt && ct && K && k && x_new && xlo && xhi && ulo && uhi && flag, "null pointer");
    const int rc = plan_check_launch(n, m, T, At, Bt, ct, K, k, x_new, xlo, xhi, ulo, uhi, flag, nullptr, nullptr, 0, nullptr,
                                     static_cast<hipStream_t>(stream));
    if (rc != IRS_OK) return rc;
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

size_t irs_iterate_scratch_bytes(int model, int mode, int T, int N) {
    int n, m, np;
    if (irs_model_info(model, &n, &m, &np) != IRS_OK || T <= 0) return 0;
    const int P = mode == IRS_ITERATE_EXACT ? 0 : irs_sums_len(model, mode);
    if (P < 0) return 0;
    size_t doubles = (size_t)T * (n * n + n * m + n + m * n + m + (size_t)P) + 8;
    size_t bytes = doubles * sizeof(double) + ((size_t)T + 16) * sizeof(int);
    bytes = (bytes + 255) / 256 * 256;
    if (mode != IRS_ITERATE_EXACT) bytes += irs_smooth_workspace_bytes(model, mode, T, N);
    return bytes;
}

int irs_iterate(const irs_iterate_call* c, irs_timing* timing, void* stream) {
    IRS_CHECK_ARG(c != nullptr, "null call struct");
    IRS_CHECK_ARG(c->T > 0 && c->n_descents > 0, "T and n_descents must be positive");
    IRS_CHECK_ARG(c->mode == IRS_ITERATE_EXACT || (c->mode >= 0 && c->mode <= 2), "unknown linearisation mode");
    IRS_CHECK_ARG(c->Q && c->Qd && c->R && c->xd_trj && c->x_trj0 && c->u_trj0, "null problem pointer");
    IRS_CHECK_ARG(c->x_hist && c->u_hist && c->cost_hist && c->info_hist && c->scratch, "null output / scratch pointer");
    int n, m, np;
    int rc = irs_model_info(c->model, &n, &m, &np);
    if (rc != IRS_OK) return rc;
    const bool exact = c->mode == IRS_ITERATE_EXACT;
    IRS_CHECK_ARG(exact || (c->N > 0 && c->std_u != nullptr), "sampled modes need N and the per-descent std_u rows");
    const size_t need = irs_iterate_scratch_bytes(c->model, c->mode, c->T, c->N);
    if (c->scratch_bytes < need) {
        irs_set_error("irs_iterate: scratch %zu < %zu bytes", c->scratch_bytes, need);
        return IRS_ERR_WORKSPACE;
    }
    const bool bounded = c->xlo && c->xhi && c->ulo && c->uhi;
    IRS_CHECK_ARG(bounded || (!c->xlo && !c->xhi && !c->ulo && !c->uhi), "give all four bound vectors or none");
    const int T = c->T;
    // carve the scratch
    double* p = static_cast<double*>(c->scratch);
    double* At = p; p += (size_t)T * n * n;
    double* Bt = p; p += (size_t)T * n * m;
    double* ct = p; p += (size_t)T * n;
    double* K = p; p += (size_t)T * m * n;
    double* k = p; p += (size_t)T * m;
    const int P = exact ? 0 : irs_sums_len(c->model, c->mode);
    double* sums = p; p += (size_t)T * P;
    p += 8;
    int* ip = reinterpret_cast<int*>(p);
    int* smooth_info = ip; ip += T;
    int* descent_info = ip; ip += 4;
    int* box_flag = ip; ip += 4;
    char* ws = static_cast<char*>(c->scratch) + (need - (exact ? 0 : irs_smooth_workspace_bytes(c->model, c->mode, T, c->N)));
    const size_t ws_bytes = exact ? 0 : irs_smooth_workspace_bytes(c->model, c->mode, T, c->N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!exact) {
        rc = irs_workspace_init(ws, ws_bytes, stream);
        if (rc != IRS_OK) return rc;
    }
    const bool box_fits = bounded && irs_tvlqr_box_lds_bytes(c->model, T) > 0 &&
                          irs_tvlqr_box_lds_bytes(c->model, T) <= (size_t)(160 * 1024 - 512);
    PhaseTimer tm(timing != nullptr, st);
    const size_t xs = (size_t)(T + 1) * n, us = (size_t)T * m;
    for (int it = 0; it < c->n_descents; ++it) {
        const double* x_nom = it == 0 ? c->x_trj0 : c->x_hist + (size_t)(it - 1) * xs;
        const double* u_nom = it == 0 ? c->u_trj0 : c->u_hist + (size_t)(it - 1) * us;
        double* x_new = c->x_hist + (size_t)it * xs;
        double* u_new = c->u_hist + (size_t)it * us;
        tm.mark(0);
        if (exact) {
            rc = irs_exact_linearize(c->model, c->params, c->n_params, T, x_nom, u_nom, At, Bt, ct, stream);
        } else {
            rc = irs_smooth_rng(c->model, c->params, c->n_params, c->mode, T, c->N, x_nom, u_nom,
                                c->std_x ? c->std_x + (size_t)it * n : nullptr, c->std_u + (size_t)it * m, c->seed,
                                c->iter0 + (uint32_t)it, sums, At, Bt, ct, smooth_info, ws, ws_bytes, stream);
        }
        if (rc != IRS_OK) return rc;
        tm.mark(1);
        // (no bounds: the descent's own launch writes the row of the info history)
        rc = irs_tvlqr_descent_row(c->model, c->params, c->n_params, T, At, Bt, ct, c->Q, c->Qd, c->R, c->alpha_R,
                                   c->xd_trj, x_nom, K, k, x_new, u_new, c->cost_hist + it, descent_info,
                                   exact ? nullptr : smooth_info, bounded ? nullptr : c->info_hist + (size_t)it * 8, stream);
        if (rc != IRS_OK) return rc;
        tm.mark(2);
        if (bounded) {
            rc = plan_check_launch(n, m, T, At, Bt, ct, K, k, x_new, c->xlo, c->xhi, c->ulo, c->uhi, box_flag, descent_info,
                                   exact ? nullptr : smooth_info, box_fits ? 0 : 1, c->info_hist + (size_t)it * 8, st);
            if (rc != IRS_OK) return rc;
            if (box_fits) {
                rc = irs_tvlqr_box_descent_if(c->model, c->params, c->n_params, T, At, Bt, ct, c->Q, c->Qd, c->R,
                                              c->alpha_R, c->xd_trj, x_nom, c->xlo, c->xhi, c->ulo, c->uhi,
                                              c->qp_rho > 0 ? c->qp_rho : 10.0, c->qp_relax > 0 ? c->qp_relax : 1.6,
                                              c->qp_max_iter > 0 ? c->qp_max_iter : 5000, c->qp_eps > 0 ? c->qp_eps : 1e-8,
                                              x_new, u_new, c->cost_hist + it, c->info_hist + (size_t)it * 8 + 3, box_flag,
                                              stream);       // its info lands in the row (zeroed above) if it runs
                if (rc != IRS_OK) return rc;
            }
        }
        tm.mark(3);
        tm.collect();
    }
    IRS_CHECK_LAUNCH();
    if (timing != nullptr) {
        timing->linearise_ms = tm.ms[0];
        timing->descent_ms = tm.ms[1];
        timing->bounds_ms = tm.ms[2];
        timing->descents = c->n_descents;
        timing->sample_steps = exact ? 0.0 : (double)c->n_descents * T * c->N;
        int d = n + m;
        bool u_only = c->mode == IRS_SMOOTH_ZERO_ORDER_B;
        timing->sample_bytes = exact ? 0.0 : timing->sample_steps * 4.0 * (u_only ? m : d);
    }
    return IRS_OK;
}

}  // extern "C"
