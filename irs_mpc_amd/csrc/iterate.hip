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
    const int block = T < 256 ? ((T + 63) / 64 * 64) : 256;
    hipLaunchKernelGGL(plan_check_kernel, dim3(1), dim3(block), 0, static_cast<hipStream_t>(stream), n, m, T, At, Bt,
                       ct, K, k, x_new, xlo, xhi, ulo, uhi, flag, nullptr, nullptr, 0, nullptr);
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
            const int block = T < 256 ? ((T + 63) / 64 * 64) : 256;
            hipLaunchKernelGGL(plan_check_kernel, dim3(1), dim3(block), 0, st, n, m, T, At, Bt, ct, K, k, x_new, c->xlo,
                               c->xhi, c->ulo, c->uhi, box_flag, descent_info, exact ? nullptr : smooth_info,
                               box_fits ? 0 : 1, c->info_hist + (size_t)it * 8);
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
