// Quasistatic descent with ONE control box on the matrix cores -- the fast form of ctrlbox.hip.
//
// Same problem, same active-set method (ctrlbox.hip has the derivation: IrsLqrQuasistatic.local_descent,
// irs_lqr/irs_lqr_quasistatic.py:286-345, re-solving solve_tvlqr's tail QP, irs_lqr/tv_lqr.py:30-137, for
// every t, as a control-box LQR in s = [x; w], w = u_{t-1}; primal-dual active-set iterations, then the
// primal active-set method), same answers -- the solution of a strictly convex QP is unique.  What differs
// is how one iteration is carried out.  Everything rides in ONE 16 x 16 tile of homogeneous coordinates
//
//     z = [s (NS); 1; (pad to NHP); nu (M)],   NHP + M <= 16
//     F_t = [A~ | B~] (NH x 16),  A~ = [A_ c_; 0 1],  B~ = [B_; 0]          s~+ = F_t z
//     L_t = stage cost as a quadratic form in z (Q~_t with the -Qs sd_t column, Ru, the cross term Nc)
//
// BACKWARD step (8 dependent v_mfma_f64_16x16x4_f64, no data movement between them):
//     Theta = L_t + F' P~ F            the Q-function of step t                      (2 x KA MFMAs)
//     K~ (M x NH): free rows -H_ff^-1 (Theta_f,s~ + H_fp b_p e_h'), pinned rows b_j e_h'   (masked LDL', lanes)
//     D  = Theta + Theta[:, nu] K~     rows < NH: Z; rows of nu: Y = Theta_nu,s~ + H K~  (1 MFMA)
//     P~ = Z + K~' Y = [I; K~]' Theta [I; K~]                                          (1 MFMA)
// -- the policy-evaluation form, valid for any K~, so rounding in K~ costs second order only; Y's pinned
// rows are the multiplier rows (mu_j = Y_j s~).  The C/D register layout of that instruction is its
// B-operand layout and the A-operand layout of the transpose, so P~ (symmetric), F' and K~' feed the next
// product as they stand (tvlqr.hip, riccati_backward_mfma, has the layout).
// FORWARD step (3 dependent MFMAs): the closed loop A~cl = A~ + B~ K~ and the M output rows (K~_j for a free
// component, Y_j for a pinned one) are stored as ONE 16 x NHP tile G_t, in A-operand register image;
//     [s~_{t+1}; out] = G_t s~_t
// yields the next state (rows < NH: already in B-operand layout for the next step) and the controls /
// multipliers (rows NHP..) together.  The state never leaves the registers.
//
// Per-step records (G_t image, P~_t, active set, bounds, iterates) live in LDS when the horizon fits
// (planar hand: T <= 53; box pivoting: T <= 123) and otherwise in a caller-supplied global workspace
// (L2-resident; same code, slower) -- the reference has no horizon limit (irs_lqr_quasistatic.py:325-345).
// f64 matrix and vector rates are equal on gfx950: what the tile buys is not flops but the absence of
// LDS round trips and cross-lane traffic inside a step (ctrlbox.hip: 6 LDS phases, ~7000 cycles per
// backward step; here ~2300 measured with in-kernel cycle stamps: 9 MFMAs at 64-108 cycles each in a
// dependent chain, ~700 cycles to gather H into every lane, ~200 for the masked inverse).
#include "boxqp.hpp"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
// the caller's (A, B, c) are read through GLOBAL-address-space pointers: a generic pointer makes them
// flat loads, which count against lgkmcnt too -- every wait for an LDS read would then also wait for the
// prefetch of the next step's data (measured: 1.8 us per backward step instead of ~0.5)
typedef const __attribute__((address_space(1))) double* gptr_d;

constexpr int kPdasIterM = 10;
// Primal-dual iterations that did not settle in kPdasIterM rounds are CYCLING (rate-limited problems: a component
// released from one bound shoots past the other one in the next round, and a whole bang-bang stretch flips with it).
// Before falling back to the primal method -- one constraint per iteration: 50-140 iterations for such a tail, each a
// sweep of up to the whole horizon; 3 of 80 tails like that were two thirds of a box-pivoting descent -- the
// iteration goes on with a damped rule: every violated bound is still pinned at once, but only the ONE pinned
// component with the worst multiplier is released per round.  Measured on the benchmark's box-pivoting loop (oracle
// twin on inputs dumped from the device, tests/tools/pdas_study.py): backward steps of the descents that used to fall
// back 13 641 -> 4 880, 10 210 -> 1 900, 24 122 -> 15 150; descents that never cycled are unchanged.  The QP is
// strictly convex: whichever rule finds the optimal set finds the same solution.
constexpr int kPdasSingleM = 50;
#ifndef IRS_LAZY_MIN
#define IRS_LAZY_MIN 6
#endif
constexpr int kLazyPrefixMin = IRS_LAZY_MIN;   // shortest fully pinned head that is left out of the inner sweeps
constexpr int KIND_ABS_M = 0, KIND_REL_M = 1;

// 1/d: hardware estimate + two Newton steps (~1 ulp).  (The policy-evaluation form would forgive a cruder
// gain in the cost-to-go, but the gain IS the control that is applied.)
__device__ __forceinline__ double fast_rcp_m(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// lane N of every 16-lane row to all lanes of that row (DPP row_newbcast; checked on gfx950: tools/microbench)
template <int N>
__device__ __forceinline__ double row_newbcast_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (bound_ctrl set: every lane has a valid source, and with it the compiler need not initialise the destination --
    // it emitted a v_mov of zero per word and broadcast otherwise, 22 per backward step)
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wmax_d(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ double wmin_d(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmin(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ int wmax_i(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = max(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ int wmin_i(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = min(v, __shfl_xor(v, s, 64));
    return v;
}

template <int NR, int M>
struct MfLayout {
    static constexpr int NS = NR + M, NH = NS + 1, NHP = (NH + 3) / 4 * 4, KA = NHP / 4;
    static constexpr bool FITS = NHP + M <= 16 && M <= 4;
    // doubles per time step: forward tile image, P~ image (lanes with col < NHP), 7 control vectors, -Qs sd
    // (the forward tile keeps its rows i < NHP + M only: index (4 s + k-group) * GR + i)
    static constexpr int GR = NHP + M, GT = KA * 4 * GR, PT = KA * 4 * NHP;
    static __host__ __device__ size_t oG(int) { return 0; }
    static __host__ __device__ size_t oP(int T) { return (size_t)T * GT; }
    static __host__ __device__ size_t oV(int T) { return oP(T) + (size_t)(T + 1) * PT; }       // 7 x (T, M)
    static __host__ __device__ size_t oQ(int T) { return oV(T) + (size_t)7 * T * M; }          // (T+1, NR)
    static __host__ __device__ size_t oS(int T) { return oQ(T) + (size_t)(T + 1) * NR; }       // (T) tile signatures
    static __host__ __device__ size_t rec_doubles(int T) { return oS(T) + (size_t)T; }
    // always in LDS: Qsym, Qdsym (NR^2), Rsym (M^2), sstart (NH), uctl (M), slack, and 128 doubles the lanes that
    // have nothing to write aim their stores at (forward sweep: unconditional stores, no exec-mask branch)
    static constexpr int small = 2 * NR * NR + M * M + NH + M + 16 + 128;
};

// TWO waves, two roles.  Wave 0 solves the tail QPs (everything below up to the MPC loop); wave 1 is the
// PLANT: it owns the realised state, applies each tail's first control to the TRUE (contact) dynamics,
// accumulates IrsLqrQuasistatic.eval_cost and hands the next start state back -- through two LDS vectors and
// two workgroup barriers per tail.  The f64 contact step needs hundreds of registers; as a separate WAVE of
// the same kernel (rather than an out-of-line FUNCTION called from the solver wave, as it used to be) it
// cannot disturb the solver's registers: no call, no value live across one.  That is the fix of the
// instantiation-dependent miscompute of the earlier kernels (wrong accumulated cost, `info`
// never written, and -- once the solver kept per-lane pointers across the call -- wrong trajectories and a
// GPU fault, all on the box-pivoting functor only): every symptom was a value of the CALLER that was live
// across the call to the contact step, whose callee (compiled with inter-procedural register allocation,
// 256 VGPRs + 222 AGPRs, no callee-saved registers) left the caller ~40 registers and a scratch frame to
// park ~150 values in.
template <class Model, int KIND, bool LDSREC>
__global__ __launch_bounds__(128) void ctrlbox_mfma_kernel(BoxArgs a, double* gws, long long* stamps) {
    constexpr int NR = Model::NX, M = Model::NU;
    using L = MfLayout<NR, M>;
    constexpr int NS = L::NS, NH = L::NH, NHP = L::NHP, KA = L::KA, RN = L::KA;   // RN: register of rows NHP..NHP+3
    static_assert(L::FITS, "one 16 x 16 tile");
    constexpr double INF = __builtin_huge_val();
    extern __shared__ double lds[];
    const int T = a.T, lane = threadIdx.x & 63, col = lane & 15, rg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // 0 solver, 1 plant
    const double tol_ = a.eps;
    // diagnostic build only (-DIRS_CBM_STAMPS, tools/stamp_descent.sh): cycle totals per phase of the solver
    // wave -> stamps[]; in the product build no stamp executes (stamps == nullptr, code compiled out)
#ifdef IRS_CBM_STAMPS
    long long st_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CBM_MARK(k) do { const long long st_now = __builtin_amdgcn_s_memtime(); st_acc[k] += st_now - st_mark; st_mark = st_now; } while (0)
#define CBM_T0() const long long st_t0 = __builtin_amdgcn_s_memtime()
#define CBM_ADD(k, n) do { st_acc[k] += __builtin_amdgcn_s_memtime() - st_t0; st_acc[(k) + 1] += (n); } while (0)
#else
#define CBM_T0() do {} while (0)
#define CBM_MARK(k) do {} while (0)
#define CBM_ADD(k, n) do {} while (0)
#endif

    double* rec = LDSREC ? lds : gws;                       // per-step records
    double* sm = LDSREC ? lds + L::rec_doubles(T) : lds;    // small tables, always LDS
    double* Gt = rec + L::oG(T);
    double* Pt = rec + L::oP(T);
    double* Vv = rec + L::oV(T);
    double* act_ = Vv;
    double* lo_ = Vv + (size_t)T * M;
    double* hi_ = Vv + (size_t)2 * T * M;
    double* uu_ = Vv + (size_t)3 * T * M;
    double* us_ = Vv + (size_t)4 * T * M;
    double* mu_ = Vv + (size_t)5 * T * M;
    double* bnd_ = Vv + (size_t)6 * T * M;                 // the bound a pinned component sits at (follows act_)
    double* qsd = rec + L::oQ(T);
    double* sig_ = rec + L::oS(T);                          // 1: step t is fully pinned and its tile was computed for this set
    double* Qsym = sm;
    double* Qdsym = Qsym + NR * NR;
    double* Rsym = Qdsym + NR * NR;
    double* sstart = Rsym + M * M;                          // NH entries: s, then 1   (plant -> solver)
    double* uctl = sstart + NH;                             // M: the tail's first control, clipped (solver -> plant)
    double* junk = uctl + M + 16;                           // 128 doubles nobody reads

    // orders this wave's memory traffic on the records: LDS executes one wave's operations in issue order
    // (compiler barrier only); global records additionally need the stores drained and this CU's L1 dropped
    auto rsync = [&]() {
        if constexpr (LDSREC) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        __builtin_amdgcn_wave_barrier();
    };

    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- the plant wave ------------------------------------------------------------------
    // barriers: S0 (the solver's tables are up), then per tail A (start state ready) and B (control ready)
    if (wave == 1) {
        double xr[NR], ur[M], xn[NR], up[M], ub[M];
        // sentinel, like info (ctrlbox.hip); stored as a bit pattern: this file is compiled with -fno-honor-nans
        if (lane == 0 && a.cost) *reinterpret_cast<unsigned long long*>(a.cost) = 0x7ff8000000000000ull;
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
        // start state [x; x[idx]; 1]: each tail's first du is measured from the realised actuated position
        // (tv_lqr.py:99-100 at the tail's local t = 0)
        auto publish_start = [&]() {
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double v = xr[0];
#pragma unroll
                for (int i = 1; i < NR; ++i) v = (i == Model::u_into_x(j)) ? xr[i] : v;
                ub[j] = v;
            }
            if (lane < NH) {
                double v = xr[0];
#pragma unroll
                for (int i = 1; i < NR; ++i) v = (i == lane) ? xr[i] : v;
#pragma unroll
                for (int j = 0; j < M; ++j) v = (NR + j == lane) ? ub[j] : v;
                sstart[lane] = lane == NS ? 1.0 : v;
            }
        };
        double cost = 0.0;
        unsigned warm = ~0u;                                // active set of the previous contact step
        irs_step_prepared<Model> pre;
        wg_barrier();                                       // S0
        publish_start();
        // The solver wave waits for this wave at A(tau+1): between B(tau) and that barrier stands only what the next
        // start state needs (the u-dependent rest of the contact step).  The bookkeeping of step tau -- its share of
        // IrsLqrQuasistatic.eval_cost (irs_lqr_quasistatic.py:153-194), the stores of x_new / u_new -- and the
        // state-only part of step tau+1 happen after A(tau+1), while the solver works.
        double xs[NR], dv[M];
        auto book = [&](int tau) {
            double e[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) e[i] = xs[i] - a.xd[(size_t)tau * NR + i];
            cost += quad(Qsym, e, NR) + quad(Rsym, dv, M);
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < M; ++j) a.u_new[(size_t)tau * M + j] = up[j];
#pragma unroll
                for (int i = 0; i < NR; ++i) a.x_new[(size_t)(tau + 1) * NR + i] = xr[i];
            }
        };
        for (int tau = 0; tau < T; ++tau) {
            wg_barrier();                                   // A(tau)
            if (tau > 0) book(tau - 1);
            // everything of the coming contact step that depends on the state alone (contact_models.hpp)
            irs_step_along_prepare<Model>(a.p, xr, warm, pre);
            wg_barrier();                                   // B(tau): uctl holds the tail's first control
#pragma unroll
            for (int j = 0; j < M; ++j) ur[j] = KIND == KIND_ABS_M ? uctl[j] : ub[j] + uctl[j];
#pragma unroll
            for (int j = 0; j < M; ++j) dv[j] = ur[j] - (tau == 0 ? ub[j] : up[j]);
#pragma unroll
            for (int i = 0; i < NR; ++i) xs[i] = xr[i];
            irs_step_along_finish<Model>(a.p, xr, ur, pre, xn, &warm);
#pragma unroll
            for (int i = 0; i < NR; ++i) xr[i] = xn[i];
#pragma unroll
            for (int j = 0; j < M; ++j) up[j] = ur[j];
            publish_start();
        }
        book(T - 1);
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
        a.info[0] = -1; a.info[1] = -1; a.info[2] = -1;     // sentinels: see ctrlbox.hip
    }
    for (int q = lane; q < NR * NR; q += 64) {
        const int i = q / NR, j = q % NR;
        Qsym[q] = 0.5 * (a.Q[i * NR + j] + a.Q[j * NR + i]);
        Qdsym[q] = 0.5 * (a.Qd[i * NR + j] + a.Qd[j * NR + i]);
    }
    for (int q = lane; q < M * M; q += 64) {
        const int i = q / M, j = q % M;
        Rsym[q] = 0.5 * (a.R[i * M + j] + a.R[j * M + i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const double* blo = KIND == KIND_ABS_M ? a.ulo : a.dlo;
    const double* bhi = KIND == KIND_ABS_M ? a.uhi : a.dhi;
    const int bs = KIND == KIND_ABS_M ? a.su : a.sd;
    for (int q = lane; q < T * M; q += 64) {
        const double lo = blo ? blo[(size_t)(q / M) * bs + q % M] : -INF;
        const double hi = bhi ? bhi[(size_t)(q / M) * bs + q % M] : INF;
        lo_[q] = lo;
        hi_[q] = hi;
        // warm start of the first tail (the previous iLQR iteration's converged set), cleaned:
        // {-1, 0, +1}, and nothing pinned at an infinite bound
        double a0 = a.act_io ? a.act_io[q] : 0.0;
        a0 = a0 < 0.0 ? (lo > -INF ? -1.0 : 0.0) : (a0 > 0.0 ? (hi < INF ? 1.0 : 0.0) : 0.0);
        act_[q] = a0;
        bnd_[q] = a0 < 0.0 ? lo : hi;
        uu_[q] = 0.0; us_[q] = 0.0; mu_[q] = 0.0;
    }
    for (int q = lane; q < (T + 1) * NR; q += 64) {
        const int t = q / NR, i = q % NR;
        double s = 0.0;
        for (int j = 0; j < NR; ++j) s += Qsym[i * NR + j] * a.xd[(size_t)t * NR + j];
        qsd[q] = -s;                                        // -(Qs sd_t)[:NR] (the tile's linear entries); the w block of sd is zero
    }
    for (size_t q = lane; q < (size_t)T * L::GT; q += 64) Gt[q] = 0.0;      // unwritten image entries stay zero
    for (int q = lane; q < T; q += 64) sig_[q] = -1.0;                        // no tile yet
    // stage-cost tile without its time-varying column, in C/D layout (row = rg + 4 r, col)
    auto Ru = [&](int i, int j) { return Rsym[i * M + j]; };
    v4d Lc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = rg + 4 * r;
        double v = 0.0;
        if (row < NR && col < NR) v = Qsym[row * NR + col];
        else if (row >= NHP && row < NHP + M && col >= NHP && col < NHP + M) v = Ru(row - NHP, col - NHP);
        if (KIND == KIND_ABS_M) {
            // (u - w)'R(u - w): R on the w block, -R between w and u
            if (row >= NR && row < NS && col >= NR && col < NS) v = Ru(row - NR, col - NR);
            else if (row >= NR && row < NS && col >= NHP && col < NHP + M) v = -Ru(row - NR, col - NHP);
            else if (col >= NR && col < NS && row >= NHP && row < NHP + M) v = -Ru(row - NHP, col - NR);
        }
        Lc[r] = v;
    }
    // terminal cost-to-go P~_T = [Qsd, -Qsd sd_T; ., 0], Qsd = diag(Qd, 0)
    if (col < NHP) {
#pragma unroll
        for (int r = 0; r < KA; ++r) {
            const int row = rg + 4 * r;
            double v = 0.0;
            if (row < NR && col < NR) v = Qdsym[row * NR + col];
            else if ((col == NS && row < NR) || (row == NS && col < NR)) {
                const int i = col == NS ? row : col;
                for (int j = 0; j < NR; ++j) v -= Qdsym[i * NR + j] * a.xd[(size_t)T * NR + j];
            }
            Pt[(size_t)T * L::PT + (r * 4 + rg) * NHP + col] = v;
        }
    }
    rsync();

    // ---- step data straight from the caller's (A, B, c): F_t in C/D layout, B~ in A-operand layout ----
    // Which array (and where) each of this lane's five elements comes from is fixed: resolved ONCE into
    // (pointer, stride per time step, constant, is-a-load), so that a step's loads are five unconditional
    // global loads (a constant element reads a.ct[0] and discards it) -- no divergent branches in the loops.
    gptr_d fp[5];
    int fstr[5];
    double fc[5];
    bool fld[5];
    {
        const gptr_d gA = (gptr_d)a.At, gB = (gptr_d)a.Bt, gc_ = (gptr_d)a.ct;
#pragma unroll
        for (int r = 0; r < 5; ++r) { fp[r] = gc_; fstr[r] = 0; fc[r] = 0.0; fld[r] = false; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rg + 4 * r;
            if (row < NR) {
                if (col < NR) { fp[r] = gA + row * NR + col; fstr[r] = NR * NR; fld[r] = true; }
                else if (col < NS) {
                    if (KIND == KIND_REL_M) { fp[r] = gB + row * M + (col - NR); fstr[r] = NR * M; fld[r] = true; }
                } else if (col == NS) { fp[r] = gc_ + row; fstr[r] = NR; fld[r] = true; }
                else if (col >= NHP && col < NHP + M) { fp[r] = gB + row * M + (col - NHP); fstr[r] = NR * M; fld[r] = true; }
            } else if (row < NS) {
                if (KIND == KIND_REL_M && col == row) fc[r] = 1.0;
                else if (col == NHP + (row - NR)) fc[r] = 1.0;
            } else if (row == NS && col == NS) {
                fc[r] = 1.0;
            }
        }
        // A-operand of B~: lane (i = col, k = rg) holds B~[i][k]
        if (rg < M) {
            if (col < NR) { fp[4] = gB + col * M + rg; fstr[4] = NR * M; fld[4] = true; }
            else if (col < NS) fc[4] = (col - NR) == rg ? 1.0 : 0.0;
        }
    }
    // raw loads now, selection (load or constant) at the point of use: the waits for the prefetched values
    // then sit where the NEXT step consumes them, not behind the loads
    auto load_F = [&](int t, double* v) {
#pragma unroll
        for (int r = 0; r < 5; ++r) v[r] = fp[r][(long)t * fstr[r]];
    };
    auto finish_F = [&](const double* v, v4d& F, double& Ba) {
#pragma unroll
        for (int r = 0; r < 4; ++r) F[r] = fld[r] ? v[r] : fc[r];
        Ba = fld[4] ? v[4] : fc[4];
    };
    // the -Qs sd_t entries of the stage-cost tile: element (row < NR, col NS) and its mirror
    int qidx[KA];
    bool qhas[KA];
#pragma unroll
    for (int r = 0; r < KA; ++r) {
        const int row = rg + 4 * r;
        qhas[r] = (col == NS && row < NR) || (row == NS && col < NR);
        qidx[r] = qhas[r] ? (col == NS ? row : col) : 0;
    }

    int bad = 0;
    // ---- backward sweep t = t_hi .. t_lo (descending): policies for the pinned sets, cost-to-go, tiles ----
    auto backward_sweep = [&](int t_hi, int t_lo) {
        if (t_hi < t_lo) return;
        CBM_T0();
        v4d P = {0.0, 0.0, 0.0, 0.0};
        if (col < NHP) {
#pragma unroll
            for (int r = 0; r < KA; ++r) P[r] = Pt[(size_t)(t_hi + 1) * L::PT + (r * 4 + rg) * NHP + col];
        }
        // the prefetched data of a step (raw (A, B, c) elements, active set, bounds, -Qs sd entries): TWO sets that
        // swap roles from step to step, two steps per loop trip -- nothing is copied at the back edge
        struct Pre { double Fraw[5]; double ac[M], bd[M], q[KA]; };
        auto load_step = [&](int t, Pre& p) {
            load_F(t, p.Fraw);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                p.ac[j] = act_[(size_t)t * M + j];
                p.bd[j] = bnd_[(size_t)t * M + j];
            }
#pragma unroll
            for (int r = 0; r < KA; ++r) p.q[r] = qsd[(size_t)t * NR + qidx[r]];
        };
        // Records in LDS: the stores of step t (P~_t, the forward tile G_t) are not issued at the end of step t, where
        // they stand between its last product and the next step's first one -- a lone wave issues in order: ~20
        // instructions, ~250 cycles on the chain for nothing -- but in the shadow of the next step's first three
        // products (P~_t IS that step's operand and stays in its registers; the tile's rows wait in `pendAcl`).
        bool pend = false;
        int pend_t = 0;
        v4d pendAcl = {0.0, 0.0, 0.0, 0.0};
        double pendGk = 0.0;
        auto flush_P = [&]() {
            double* pp = col < NHP ? Pt + (size_t)pend_t * L::PT + (size_t)rg * NHP + col : junk + lane;
            const int ps = col < NHP ? 4 * NHP : 0;
#pragma unroll
            for (int r = 0; r < KA; ++r) pp[r * ps] = P[r];
        };
        auto flush_G = [&]() {
            double* G = Gt + (size_t)pend_t * L::GT;
            double* gq = col < NH ? G + (size_t)col * L::GR + rg : junk + 64 + lane;
            const int gs = col < NH ? 4 : 0;
#pragma unroll
            for (int r = 0; r < KA; ++r) gq[r * gs] = pendAcl[r];
            double* gk = (col < NH && rg < M) ? G + (size_t)col * L::GR + NHP + rg : junk + 64 + lane;
            *gk = pendGk;
        };
        auto bw_step = [&](int t, const Pre& cur, Pre& nxt, bool prefetch) {
#ifdef IRS_CBM_STAMPS
            {   // how long the step waits for its prefetched (A, B, c)
                const long long w0 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st_acc[7] += __builtin_amdgcn_s_memtime() - w0;
            }
#endif
#ifdef IRS_CBM_STAMPS
            long long st_mark = __builtin_amdgcn_s_memtime();
#endif
            v4d F;
            double Ba;
            finish_F(cur.Fraw, F, Ba);
            // active set and pinned values of this step (wave-uniform)
            double ac[M], bb[M];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                ac[j] = cur.ac[j];
                bb[j] = cur.bd[j];
            }
            v4d Lt = Lc;
            // D1 = P~ F ;  Theta = L_t + F' D1.  What is not on the chain is DEALT OUT over the gaps between the six
            // products by hand, one scheduling barrier per gap (left to itself the compiler issues all of it in one gap:
            // ~40 instructions behind a 64-cycle product stall the chain ~190 cycles per step)
            v4d D1 = {0.0, 0.0, 0.0, 0.0};
            v4d Th;
            if constexpr (LDSREC && KA == 3) {
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0], F[0], D1, 0, 0, 0);
                if (pend) flush_P();                                             // gap 1: last step's P~ record
                __builtin_amdgcn_sched_barrier(0);
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[1], F[1], D1, 0, 0, 0);
                if (pend) flush_G();                                             // gap 2: last step's forward tile
                __builtin_amdgcn_sched_barrier(0);
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[2], F[2], D1, 0, 0, 0);
                if (prefetch) load_F(t - 1, nxt.Fraw);                           // gap 3: next step's (A, B, c) from L2
#pragma unroll
                for (int r = 0; r < KA; ++r) Lt[r] = qhas[r] ? cur.q[r] : Lc[r];
                __builtin_amdgcn_sched_barrier(0);
                Th = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], D1[0], Lt, 0, 0, 0);
                if (prefetch) {                                                  // gap 4: next step's set and bounds
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        nxt.ac[j] = act_[(size_t)(t - 1) * M + j];
                        nxt.bd[j] = bnd_[(size_t)(t - 1) * M + j];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                Th = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], D1[1], Th, 0, 0, 0);
                if (prefetch) {                                                  // gap 5: next step's -Qs sd entries
#pragma unroll
                    for (int r = 0; r < KA; ++r) nxt.q[r] = qsd[(size_t)(t - 1) * NR + qidx[r]];
                }
                __builtin_amdgcn_sched_barrier(0);
                Th = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2], D1[2], Th, 0, 0, 0);
            } else if constexpr (LDSREC && KA == 2) {
                // (the 8-row tiles: four products, three gaps)
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0], F[0], D1, 0, 0, 0);
                if (pend) flush_P();
                __builtin_amdgcn_sched_barrier(0);
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[1], F[1], D1, 0, 0, 0);
                if (pend) flush_G();
                if (prefetch) load_F(t - 1, nxt.Fraw);
#pragma unroll
                for (int r = 0; r < KA; ++r) Lt[r] = qhas[r] ? cur.q[r] : Lc[r];
                __builtin_amdgcn_sched_barrier(0);
                Th = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], D1[0], Lt, 0, 0, 0);
                if (prefetch) {
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        nxt.ac[j] = act_[(size_t)(t - 1) * M + j];
                        nxt.bd[j] = bnd_[(size_t)(t - 1) * M + j];
                    }
#pragma unroll
                    for (int r = 0; r < KA; ++r) nxt.q[r] = qsd[(size_t)(t - 1) * NR + qidx[r]];
                }
                __builtin_amdgcn_sched_barrier(0);
                Th = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], D1[1], Th, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < KA; ++r) Lt[r] = qhas[r] ? cur.q[r] : Lc[r];
                if (prefetch) load_step(t - 1, nxt);               // prefetch
                if constexpr (LDSREC) {
                    D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0], F[0], D1, 0, 0, 0);
                    if (pend) flush_P();
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (KA >= 2) D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[1], F[1], D1, 0, 0, 0);
                    if (pend) flush_G();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 2; s < KA; ++s) D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[s], F[s], D1, 0, 0, 0);
                } else {
#pragma unroll
                    for (int s = 0; s < KA; ++s) D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[s], F[s], D1, 0, 0, 0);
                }
                Th = Lt;
#pragma unroll
                for (int s = 0; s < KA; ++s) Th = __builtin_amdgcn_mfma_f64_16x16x4f64(F[s], D1[s], Th, 0, 0, 0);
            }
            // H = Theta_nu,nu to every lane (symmetric: 10 of 16 by readlane); this lane's column of
            // Theta_nu,: (rows NHP + i sit in register RN of lanes 16 i + col).
            // The active set enters as 0/1 factors mf_i (free) -- products instead of per-element selects: a
            // lone wave pays ~5 cycles for EVERY instruction, and the step is instruction-bound.
            // (tried: __builtin_amdgcn_sched_group_barrier pipelines to spread the head's ~55 selects, compares and
            // prefetch loads evenly over the gaps between the six products -- the scheduler piles them up behind the
            // last one instead, because Theta accumulates into the registers P is read from: no change, 837 cycles)
            asm volatile("" :: "v"(Th[0]), "v"(Th[RN]));
            CBM_MARK(8);                                      // head + prefetch issue + 6 MFMAs
            // A step whose m components are ALL pinned needs neither H nor its inverse: K~ = b e_h'.  (73 % of the
            // backward steps of the trust-region benchmark: the re-swept stretch of a tail is its saturated head.)
            bool allpin = true;
#pragma unroll
            for (int j = 0; j < M; ++j) allpin = allpin && ac[j] != 0.0;
            double Kb = 0.0;
            bool my_free = false;
            if (__builtin_amdgcn_readfirstlane((int)allpin)) {
#pragma unroll
                for (int i = 0; i < M; ++i)
                    if (i == rg) Kb = col == NS ? bb[i] : 0.0;
                // this tile's closed loop holds for as long as the step's active set does (lazy prefix, below):
                // sig_ = 1 here, reset to -1 wherever act_ changes
                if constexpr (LDSREC) {
                    double* sp = lane == 0 ? sig_ + t : junk + lane;
                    *sp = 1.0;
                } else if (lane == 0) {
                    sig_[t] = 1.0;
                }
                CBM_MARK(9);
            } else {
                double H[M][M], gc[M], mf[M], bz[M];
                // One product with a constant selector replicates the rows of Theta_nu over the lane groups:
                //   Gc[r] (every lane) = Theta_nu[r][col]      (A[row][k] = [k == row >> 2], B = register RN as it is)
                // -- that is gc -- and H[i][j] = Theta_nu[i][NHP + j] is lane NHP + j of each 16-lane row of Gc[i]:
                // a DPP row broadcast.  (Instead of 20 v_readlane into SGPRs + 8 ds_bpermute; staging through an LDS
                // tile was tried too.  Measured with the in-kernel stamps: whichever way H travels, gather + inverse
                // of a free step take ~1100 cycles together -- the wait for Theta's last product and then ~110 f64
                // operations of one wave at ~8 cycles each; this form keeps the SGPRs free and is 1 % faster.)
                const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
                const v4d Gc = __builtin_amdgcn_mfma_f64_16x16x4f64(rg == (col >> 2) ? 1.0 : 0.0, Th[RN], zero4, 0, 0, 0);
                auto bc = [&](double v, int j) {
                    switch (j) {
                        case 0: return row_newbcast_d<NHP + 0>(v);
                        case 1: return row_newbcast_d<(NHP + 1) & 15>(v);
                        case 2: return row_newbcast_d<(NHP + 2) & 15>(v);
                        default: return row_newbcast_d<(NHP + 3) & 15>(v);
                    }
                };
#pragma unroll
                for (int i = 0; i < M; ++i) {
#pragma unroll
                    for (int j = i; j < M; ++j) {
                        H[i][j] = bc(Gc[i], j);
                        H[j][i] = H[i][j];
                    }
                    gc[i] = Gc[i];
                    mf[i] = ac[i] == 0.0 ? 1.0 : 0.0;
                    bz[i] = ac[i] == 0.0 ? 0.0 : bb[i];          // pinned value, 0 for a free component
                }
                asm volatile("" :: "v"(H[0][0]), "v"(gc[0]), "v"(gc[M - 1]));
                CBM_MARK(9);                                      // gather of H and the lane's column
                // Masked inverse: free rows/columns of H, identity on the pinned ones -- in CLOSED FORM (2 x 2 blocks:
                // A^-1, the Schur complement S = C - B'A^-1 B, S^-1), not by an LDL' with substitutions.  A
                // dependent f64 instruction of a lone wave costs ~32 cycles, an independent one ~8
                // (tools/microbench/mfma_f64_latency.hip): the LDL' + two substitutions were a ~55-deep chain,
                // ~1900 of the step's 2440 cycles; the block form is ~22 deep with plenty to issue beside it.
                double Hm[M][M], Hi[M][M], y[M];
    #pragma unroll
                for (int i = 0; i < M; ++i)
    #pragma unroll
                    for (int j = 0; j < M; ++j) Hm[i][j] = H[i][j];
                // a step with NO pinned component (most steps of a converging iLQR run) needs no masking and no
                // pinned-value terms: ~35 of the phase's ~110 f64 operations, behind a wave-uniform branch
                bool anypin = false;
    #pragma unroll
                for (int j = 0; j < M; ++j) anypin = anypin || ac[j] != 0.0;
                const bool pins = __builtin_amdgcn_readfirstlane((int)anypin) != 0;
                if (pins) {
    #pragma unroll
                    for (int i = 0; i < M; ++i)
    #pragma unroll
                        for (int j = 0; j < M; ++j) Hm[i][j] = i == j ? (ac[i] == 0.0 ? H[i][i] : 1.0) : H[i][j] * (mf[i] * mf[j]);
                }
                const double hsel = col == NS ? 1.0 : 0.0;
                double rhs[M];
    #pragma unroll
                for (int i = 0; i < M; ++i) {
                    // right-hand side of the free rows: Theta_f,col + (homogeneous column) sum_p H_fp b_p
                    rhs[i] = gc[i];
                }
                if (pins) {
    #pragma unroll
                    for (int i = 0; i < M; ++i) {
                        double hb0 = 0.0, hb1 = 0.0;
    #pragma unroll
                        for (int l = 0; l < M; ++l) {
                            if (l & 1) hb1 = fma(H[i][l], bz[l], hb1);
                            else hb0 = fma(H[i][l], bz[l], hb0);
                        }
                        rhs[i] = fma(hsel, hb0 + hb1, gc[i]) * mf[i];
                    }
                }
                bool spd = true;
                auto inv2 = [&](double p, double q, double r, double& ip, double& iq, double& ir) {
                    // [p q; q r]^-1 = [r -q; -q p] / (p r - q^2)
                    const double det = fma(p, r, -q * q);
                    spd = spd && p > 0.0 && det > 0.0;
                    const double id = fast_rcp_m(det);
                    ip = r * id; iq = -q * id; ir = p * id;
                };
                if constexpr (M == 1) {
                    spd = Hm[0][0] > 0.0;
                    Hi[0][0] = fast_rcp_m(Hm[0][0]);
                } else if constexpr (M == 2) {
                    inv2(Hm[0][0], Hm[0][1], Hm[1][1], Hi[0][0], Hi[0][1], Hi[1][1]);
                    Hi[1][0] = Hi[0][1];
                } else {
                    static_assert(M == 1 || M == 2 || M == 4, "closed-form inverse for 1, 2 or 4 controls");
                    double a0, a1, a2;                                   // A^-1 (symmetric: a0 a1; a1 a2)
                    inv2(Hm[0][0], Hm[0][1], Hm[1][1], a0, a1, a2);
                    // W = A^-1 B (2 x 2), S = C - B' W
                    const double w00 = fma(a0, Hm[0][2], a1 * Hm[1][2]), w01 = fma(a0, Hm[0][3], a1 * Hm[1][3]);
                    const double w10 = fma(a1, Hm[0][2], a2 * Hm[1][2]), w11 = fma(a1, Hm[0][3], a2 * Hm[1][3]);
                    const double s00 = Hm[2][2] - fma(Hm[0][2], w00, Hm[1][2] * w10);
                    const double s01 = Hm[2][3] - fma(Hm[0][2], w01, Hm[1][2] * w11);
                    const double s11 = Hm[3][3] - fma(Hm[0][3], w01, Hm[1][3] * w11);
                    double c0, c1, c2;                                   // S^-1
                    inv2(s00, s01, s11, c0, c1, c2);
                    // the lane's own right-hand side is SOLVED through the blocks -- y2 = S^-1 (r2 - B' A^-1 r1),
                    // y1 = A^-1 r1 - W y2 -- instead of assembling the 4 x 4 inverse and multiplying: 16 operations after
                    // S^-1 where the inverse took 17 and its application 16
                    const double t0 = fma(a0, rhs[0], a1 * rhs[1]), t1 = fma(a1, rhs[0], a2 * rhs[1]);
                    const double q0 = rhs[2] - fma(Hm[0][2], t0, Hm[1][2] * t1);
                    const double q1 = rhs[3] - fma(Hm[0][3], t0, Hm[1][3] * t1);
                    y[2] = fma(c0, q0, c1 * q1);
                    y[3] = fma(c1, q0, c2 * q1);
                    y[0] = t0 - fma(w00, y[2], w01 * y[3]);
                    y[1] = t1 - fma(w10, y[2], w11 * y[3]);
                }
                if (!spd && bad == 0) bad = t + 1;
                if constexpr (M != 4) {
    #pragma unroll
                    for (int i = 0; i < M; ++i) {
                        double y0 = 0.0, y1 = 0.0;
    #pragma unroll
                        for (int l = 0; l < M; ++l) {
                            if (l & 1) y1 = fma(Hi[i][l], rhs[l], y1);
                            else y0 = fma(Hi[i][l], rhs[l], y0);
                        }
                        y[i] = y0 + y1;
                    }
                }
                // K~[rg][col] in B-operand layout (= K~' in A-operand layout): free rows -y, pinned rows b e_h'
                // (y = 0 on a pinned row, bz = 0 on a free one)
                double mfree = 0.0;
    #pragma unroll
                for (int i = 0; i < M; ++i) {
                    const double kv = hsel * bz[i] - y[i];
                    if (i == rg) { Kb = kv; mfree = mf[i]; }
                }
                if (rg >= M || col >= NH) Kb = 0.0;
                my_free = mfree != 0.0;
            }
            asm volatile("" :: "v"(Kb));
            CBM_MARK(10);                                     // inverse, right-hand sides, gains
            // D = Theta + Theta[:, nu] K~ ;  P~ = D + K~' Y, Y = rows of nu of D
            v4d D = __builtin_amdgcn_mfma_f64_16x16x4f64(Th[RN], Kb, Th, 0, 0, 0);
            v4d Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(Kb, D[RN], D, 0, 0, 0);
            // closed loop A~cl = A~ + B~ K~ (off the chain of the next step)
            v4d Acl = __builtin_amdgcn_mfma_f64_16x16x4f64(Ba, Kb, F, 0, 0, 0);
            // P~ for the next step.  Entries outside NH x NH (the nu rows and columns of the tile) are finite leftovers;
            // as the A operand of the next step they only reach output rows >= NH, which nothing reads -- so no masking
            // when the tile has no padding between s~ and nu (NH == NHP: both shipped models)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rg + 4 * r;
                if constexpr (NH == NHP) P[r] = r < KA ? Pn[r] : 0.0;
                else P[r] = (r < KA && row < NH && col < NH) ? Pn[r] : 0.0;
            }
            // records: P~_t, and the forward tile G_t as its A-operand will be read -- element [i][k] belongs to
            // register k >> 2 of lane (k & 3) * 16 + i, stored at k * GR + i.  Records in LDS: every lane stores, the
            // lanes outside the tile into `junk` (no exec-mask branches: see the forward sweep)
            double* G = Gt + (size_t)t * L::GT;
            if constexpr (LDSREC) {
                pend = true;
                pend_t = t;
                pendAcl = Acl;
                pendGk = my_free ? Kb : D[RN];
            } else {
                if (col < NHP) {
#pragma unroll
                    for (int r = 0; r < KA; ++r) Pt[(size_t)t * L::PT + (r * 4 + rg) * NHP + col] = P[r];
                }
                if (col < NH) {
#pragma unroll
                    for (int r = 0; r < KA; ++r) G[col * L::GR + (rg + 4 * r)] = Acl[r];
                    if (rg < M) G[col * L::GR + (NHP + rg)] = my_free ? Kb : D[RN];
                }
            }
            asm volatile("" :: "v"(P[0]));
            CBM_MARK(11);                                     // 3 MFMAs, masks, record stores
        
        };
        Pre pa, pb;
        load_step(t_hi, pa);
        int t = t_hi;
        for (; t - 1 >= t_lo; t -= 2) {
            bw_step(t, pa, pb, true);
            bw_step(t - 1, pb, pa, t - 2 >= t_lo);
        }
        if (t >= t_lo) bw_step(t, pa, pb, false);
        if constexpr (LDSREC) {
            if (pend) {
                flush_P();
                flush_G();
            }
        }
        rsync();
        CBM_ADD(0, t_hi - t_lo + 1);
    };

    // ---- policy rollout on the linear model from sstart: controls -> `dst` (uu_ or us_), multipliers -> mu_ ----
    // `clip` (trust-region problems, phase 1): the controls of FREE components are clipped to their boxes as the rollout
    // goes -- s~+ = G s~ + B~ (clip(nu) - nu), a fourth product per step -- so that the violations the round reports are
    // those of a trajectory the plant could follow, not of one that has already left the box upstream.  It is only a
    // better update rule (the accepted round has nothing to clip: identical to the plain rollout, and every condition
    // is still verified on it): on the benchmark's 20 planar-hand descents 24 % fewer backward steps and 12 % fewer
    // forward steps (oracle twin on dumped inputs, tests/tools/pdas_study.py: 20 694 -> 15 609 / 52 381 -> 46 055);
    // rate-limited problems get WORSE with it (23 984 -> 44 179 on four box-pivoting descents) and do not use it.
    auto policy_rollout = [&](int t0, double* dst, int te, bool clip = false) {          // steps t0 .. te-1
        CBM_T0();
        v4d S = {0.0, 0.0, 0.0, 0.0};
        if (col == 0) {
#pragma unroll
            for (int s = 0; s < KA; ++s) {
                const int k = rg + 4 * s;
                S[s] = k < NH ? sstart[k] : 0.0;
            }
        }
        const int gofs = col < L::GR ? col : 0;
        const bool out_lane = col == 0 && rg < M;
        const int cj = rg < M ? rg : 0;
        // What a step costs is its instruction count, not the latency of its three products: a lone wave issues in
        // order, so only what stands BETWEEN two dependent matrix instructions runs in their shadow (64 cycles
        // each), and an LDS or f64-compare instruction costs ~10 cycles there (measured: the chain alone 216 cycles
        // per step, tools/microbench; the former loop body -- 6 reads, 2 compares, 8 selects, 2 masked writes, address
        // adds -- 480).  The loop therefore only loads the tile and stores the RAW output row (register RN: K~_j s~
        // of a free component, the multiplier Y_j s~ of a pinned one) into mu_; controls and multipliers are sorted
        // out afterwards, four (t, j) entries per lane at once.
        struct Fw { double g[KA]; double ti, lo, hi, ac; };
        // steps ta .. tb-1 from state S (returned in S); `clipping`: free controls clipped as the rollout goes -- the
        // three bound reads and the ~10 dependent f64 instructions between two steps' products make that step 520
        // cycles instead of 320, which is why the caller below only turns it on where something does clip
        auto roll_range = [&](int ta, int tb, v4d& S, bool clipping) {
            const double* gp = Gt + (size_t)ta * L::GT + (size_t)rg * L::GR + gofs;
            int tq = ta;
            auto fetch = [&](Fw& f) {
#pragma unroll
                for (int s = 0; s < KA; ++s) f.g[s] = gp[4 * s * L::GR];
                gp += L::GT;
                if (clipping) {
                    f.ti = (double)tq;              // the step's index: its element of B~ is fetched only if something clips
                    f.lo = lo_[(size_t)tq * M + cj];
                    f.hi = hi_[(size_t)tq * M + cj];
                    f.ac = act_[(size_t)tq * M + cj];
                    ++tq;
                }
            };
            // every lane stores (records in LDS): lanes without an output aim at `junk`, stride 0 -- a store under an
            // exec-mask branch would make the count of outstanding LDS operations unknown to the compiler
            double* mo = (out_lane || !LDSREC) ? mu_ + (size_t)ta * M + (rg < M ? rg : 0) : junk + lane;
            const int mstride = (out_lane || !LDSREC) ? M : 0;
            auto emit = [&](const v4d& Dp) {
                if constexpr (LDSREC) *mo = Dp[RN];
                else if (out_lane) *mo = Dp[RN];
                mo += mstride;
            };
            auto step = [&](const Fw& f, const v4d& Sin, v4d& Sout, bool has_prev) {
                v4d Dn = {0.0, 0.0, 0.0, 0.0};
                Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(col < L::GR ? f.g[0] : 0.0, Sin[0], Dn, 0, 0, 0);
                if (has_prev) emit(Sin);                           // the previous step's outputs, in this product's shadow
#pragma unroll
                for (int s = 1; s < KA; ++s)
                    Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(col < L::GR ? f.g[s] : 0.0, Sin[s], Dn, 0, 0, 0);
                if (clipping) {
                    // register RN of lane (col 0, rg) is nu_rg of this step (the multiplier for a pinned component: left
                    // alone) and, as the fourth k-group of the B operand, the place of delta_rg
                    const double raw = Dn[RN];
                    const double cl = fmin(fmax(raw, f.lo), f.hi);
                    const double dlt = (out_lane && f.ac == 0.0) ? cl - raw : 0.0;
                    // (the product only where some lane has a delta; B~ comes from L2, a microsecond away: prefetching
                    // it every step cost 200 cycles per step)
                    if (__ballot(dlt != 0.0) != 0ull) {
                        const double bav = fp[4][(long)(int)f.ti * fstr[4]];
                        Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(fld[4] ? bav : fc[4], dlt, Dn, 0, 0, 0);
                    }
                }
                Sout = Dn;
            };
            // two steps per trip: the prefetched operands and the state tile alternate between two register sets,
            // so nothing is copied at the back edge (a lone wave pays ~5 cycles for every v_mov)
            Fw fa, fb;
            v4d S1;
            fetch(fa);
            int t = ta;
            bool hp = false;
            for (; t + 1 < tb; t += 2) {
                fetch(fb);
                step(fa, S, S1, hp);
                if (t + 2 < tb) fetch(fa);
                step(fb, S1, S, true);
                hp = true;
            }
            if (t < tb) {
                step(fa, S, S1, hp);
                emit(S1);
                S = S1;
            } else if (hp) {
                emit(S);
            }
        };
#ifdef IRS_CBM_STAMPS
        const long long st_fl0 = __builtin_amdgcn_s_memtime();
#endif
        if (!clip) {
            roll_range(t0, te, S, false);
        } else {
            // plain steps, kCheck at a time, until a free control leaves its box inside a chunk: that chunk is redone, and
            // the rest done, with clipping.  (The accepted round of every tail, and the head of most others, clips nothing.)
            #ifndef IRS_KCHECK
#define IRS_KCHECK 16
#endif
            constexpr int kCheck = IRS_KCHECK;
            int t = t0;
            bool found = false;
            while (t < te && !found) {
                const int tc = min(t + kCheck, te);
                const v4d Ssave = S;
                roll_range(t, tc, S, false);
#ifdef IRS_CBM_STAMPS
                st_acc[20] += tc - t;
                st_acc[21] += 1;
#endif
                rsync();
                bool v = false;
                for (int q = t * M + lane; q < tc * M; q += 64) {
                    const double raw = mu_[q];
                    v = v || (act_[q] == 0.0 && (raw < lo_[q] || raw > hi_[q]));
                }
                if (__any(v)) {
                    S = Ssave;
                    found = true;
                } else {
                    t = tc;
                }
            }
#ifdef IRS_CBM_STAMPS
            st_acc[19] += te - t;
#endif
            if (t < te) roll_range(t, te, S, true);
        }
#ifdef IRS_CBM_STAMPS
        asm volatile("" :: "v"(S[0]));
        st_acc[13] += __builtin_amdgcn_s_memtime() - st_fl0;
        st_acc[12] += 1;
#endif
        rsync();
        // controls and multipliers from the raw rows
        for (int q = t0 * M + lane; q < te * M; q += 64) {
            const double raw = mu_[q], ac = act_[q];
            const double bd = bnd_[q];
            dst[q] = ac == 0.0 ? raw : bd;
            mu_[q] = ac == 0.0 ? 0.0 : raw;
        }
        rsync();
        CBM_ADD(2, te - t0);
    };

    // ---- cold start of the first tail: the SATURATED unconstrained policy.  With every component free the
    // sweep above is the plain Riccati pass; rolling its policy out on the linear model with the controls
    // clipped to their boxes (s~+ = A~cl s~ + B~ (clip(nu) - nu)) pins what that rollout saturates -- most of
    // the set the QP's solution binds.  From there the active-set iterations of the benchmark's first tail
    // sweep ~230 time steps instead of ~1090 (oracle twin: local_descent_quasistatic_as, sat_start).  The
    // starting set does not change the answer (strictly convex QP); once per descent, T steps.
    auto saturated_start = [&](int t0) {
        v4d S = {0.0, 0.0, 0.0, 0.0};
        if (col == 0) {
#pragma unroll
            for (int s = 0; s < KA; ++s) {
                const int k = rg + 4 * s;
                S[s] = k < NH ? sstart[k] : 0.0;
            }
        }
        const gptr_d gB = (gptr_d)a.Bt;
        for (int t = t0; t < T; ++t) {
            const double* G = Gt + (size_t)t * L::GT;
            v4d Dn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KA; ++s) {
                const double g = col < L::GR ? G[(4 * s + rg) * L::GR + (col < L::GR ? col : 0)] : 0.0;
                Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(g, S[s], Dn, 0, 0, 0);
            }
            // unclipped controls sit in register RN of lanes 16 j: clip, record the pin, broadcast the excess
            double dlt[M];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const double v = readlane_d(Dn[RN], 16 * j);
                const double lo = lo_[(size_t)t * M + j], hi = hi_[(size_t)t * M + j];
                const double c = fmin(fmax(v, lo), hi);
                dlt[j] = c - v;
                if (lane == 0) {
                    act_[(size_t)t * M + j] = v < lo - tol_ ? -1.0 : (v > hi + tol_ ? 1.0 : 0.0);
                    bnd_[(size_t)t * M + j] = v < lo - tol_ ? lo : hi;
                    sig_[t] = -1.0;
                }
            }
            // s~+ rows (column 0): x rows += B delta, w rows += delta
            if (col == 0) {
#pragma unroll
                for (int s = 0; s < KA; ++s) {
                    const int row = rg + 4 * s;
                    double corr = 0.0;
                    if (row < NR) {
#pragma unroll
                        for (int j = 0; j < M; ++j) corr += gB[((size_t)t * NR + row) * M + j] * dlt[j];
                    } else if (row < NS) {
#pragma unroll
                        for (int j = 0; j < M; ++j) corr = (row - NR == j) ? dlt[j] : corr;
                    }
                    S[s] = Dn[s] + corr;
                }
            }
        }
        rsync();
    };

    // ---- MPC loop (solver side) ----------------------------------------------------------
#ifdef IRS_CBM_STAMPS
    const long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    int it_max = 0, n_fail = 0;
    const double tol = a.eps;
    bool full = true;                                  // no valid backward sweep yet
    wg_barrier();                                      // S0: tables and records are up

    for (int tau = 0; tau < T; ++tau) {
        {
            CBM_T0();
            wg_barrier();                              // A(tau): the plant has published this tail's start state
            CBM_ADD(4, 1);
        }
        const int t0 = tau;
        if (tau == 0) {
            // no warm start handed in (act_io absent or all zero): start from the saturated policy
            bool any = false;
            for (int q = lane; q < T * M; q += 64) any = any || act_[q] != 0.0;
            if (!__any(any)) {
                backward_sweep(T - 1, 0);
                saturated_start(0);
            }
        }
        int t_dirty = full ? T - 1 : t0 - 1;           // the sweep of the previous tail covers t >= tau
        int iters = 0;
        bool conv = false;
        // ---- phase 1: primal-dual active set, with a LAZY PREFIX.  A fully pinned step's closed loop is
        // A~ + B~ b e_h' whatever the cost-to-go: the rollout through it does not depend on the sweep; only its
        // multiplier rows Y_t do.  The saturated head of a trust-region tail -- the contiguous run of fully pinned
        // steps from t0 whose tiles were computed for the current set (sig_) -- is therefore left out of the inner
        // iterations' sweeps (its multipliers are not consulted meanwhile) and swept ONCE, when the rest has
        // settled: then its multipliers are tested, and a wrong-signed one re-opens the iteration.  The solution of
        // the strictly convex QP does not depend on the order in which violated conditions are repaired; every
        // condition is verified on fresh records before the tail is accepted.  (Trust-region benchmark: 73 % of the
        // backward steps were re-sweeps of that head.)
        int stale_hi = t0 - 1;                         // records of [t0, stale_hi] are out of date (skipped prefix)
        auto prefix_end = [&]() {                      // last step of the run described above (t0 - 1: none)
            int pe = t0 - 1;
            for (int base = t0; base < T; base += 64) {
                const int t = base + lane;
                bool ok = false;
                if (t < T) {
                    bool allp = sig_[t] == 1.0;
#pragma unroll
                    for (int j = 0; j < M; ++j) allp = allp && act_[(size_t)t * M + j] != 0.0;
                    ok = allp;
                }
                const unsigned long long bal = __ballot(ok);
                const int run = bal == ~0ull ? 64 : __builtin_ctzll(~bal);
                pe = base + run - 1;
                if (run < 64) break;
            }
            return pe;
        };
        for (int it = 0; it < kPdasIterM + kPdasSingleM && !conv; ++it) {
            ++iters;
            const bool single = it >= kPdasIterM;          // release the worst pinned component only
            // (a short run is not worth an extra rollout and a later release: measured on the rate-limited box
            // problem, where skipping 1-3 steps cost more iterations than it saved sweeps)
            int pe = t0 - 1;
            // Trust-region (ABS) problems only: their tails start with a long saturated head.  Rate-limited (REL)
            // ones rarely do, and deferring the few releases there costs iterations (measured on the benchmark's
            // box-pivoting loop: 206 -> 187 iterations/s with it, planar hand trust region 579 -> 596).
            if (KIND == KIND_ABS_M && max(t_dirty, stale_hi) >= t0) {   // (nothing to sweep: no need to know)
                pe = prefix_end();
                if (pe - t0 + 1 < kLazyPrefixMin) pe = t0 - 1;
            }
            {
                const int hi = max(t_dirty, stale_hi), lo = max(t0, pe + 1);
                if (hi >= lo) backward_sweep(hi, lo);
                if (hi >= t0) stale_hi = lo - 1;       // what was dirty below lo stays so
            }
            policy_rollout(t0, uu_, T, KIND == KIND_ABS_M);
            int chg = -1;
            double rworst = 0.0;
            int rq = 0x7fffffff;
            // (every lane runs every trip -- `chg` comes out of a ballot and must be the same in all of them)
            for (int qb = t0 * M; qb < T * M; qb += 64) {
                const int q = qb + lane;
                bool changed = false;
                if (q < T * M) {
                    const double ac = act_[q], u = uu_[q], mu = mu_[q];
                    const bool fresh = q / M > stale_hi;   // multipliers of the skipped prefix are not current
                    double nw = ac;
                    if (ac == 0.0) {
                        if (u < lo_[q] - tol) nw = -1.0;
                        else if (u > hi_[q] + tol) nw = 1.0;
                    } else if (ac < 0.0) {
                        if (fresh && mu < -tol) nw = 0.0;
                    } else {
                        if (fresh && mu > tol) nw = 0.0;
                    }
                    if (single && nw == 0.0 && ac != 0.0) {     // a release: remembered, not applied
                        const double v = fabs(mu);
                        if (v > rworst) { rworst = v; rq = q; }
                        nw = ac;
                    }
                    changed = nw != ac;
                    if (changed) { act_[q] = nw; bnd_[q] = nw < 0.0 ? lo_[q] : hi_[q]; sig_[q / M] = -1.0; }
                }
                // the latest changed step, without a cross-lane reduction (a wave maximum is six LDS-crossbar
                // shuffles, ~500 cycles per round): q grows with the lane, so it is the ballot's highest bit
                const unsigned long long cb = __ballot(changed);
                if (cb != 0ull) chg = max(chg, (qb + 63 - __builtin_clzll(cb)) / M);
            }
            if (single) {
                const double wm = wmax_d(rworst);
                if (wm > 0.0) {
                    const int qw = wmin_i(rworst == wm ? rq : 0x7fffffff);      // first index among equals
                    if (lane == 0) { act_[qw] = 0.0; sig_[qw / M] = -1.0; }
                    chg = max(chg, qw / M);
                }
            }
            rsync();
            if (chg < 0 && stale_hi >= t0) {
                // the rest has settled: sweep the skipped prefix once and test ITS multipliers
                const int ph = stale_hi;
#ifdef IRS_CBM_STAMPS
                st_acc[14] += ph - t0 + 1;
#endif
                backward_sweep(ph, t0);
                stale_hi = t0 - 1;
                policy_rollout(t0, uu_, ph + 1);
                for (int qb = t0 * M; qb < (ph + 1) * M; qb += 64) {
                    const int q = qb + lane;
                    bool rel = false;
                    if (q < (ph + 1) * M) {
                        const double ac = act_[q], mu = mu_[q];
                        rel = (ac < 0.0 && mu < -tol) || (ac > 0.0 && mu > tol);
                        if (rel) { act_[q] = 0.0; sig_[q / M] = -1.0; }
                    }
                    const unsigned long long cb = __ballot(rel);
                    if (cb != 0ull) chg = max(chg, (qb + 63 - __builtin_clzll(cb)) / M);
                }
                rsync();
#ifdef IRS_CBM_STAMPS
                st_acc[15] += chg >= 0 ? 1 : 0;
#endif
            }
            if (chg < 0) conv = true;
            else t_dirty = chg;
        }
        if (stale_hi >= t0) { t_dirty = max(t_dirty, stale_hi); stale_hi = t0 - 1; }      // phase 2 sweeps everything dirty
        // ---- phase 2: primal active set from the clipped iterate
        if (!conv) {
#ifdef IRS_CBM_STAMPS
            st_acc[16] += 1;
            st_acc[18] += iters;
#endif
            int chg = -1;
            for (int q = t0 * M + lane; q < T * M; q += 64) {
                const double lo = lo_[q], hi = hi_[q];
                const double u = fmin(fmax(uu_[q], lo), hi);
                uu_[q] = u;
                const double nw = u <= lo ? -1.0 : (u >= hi ? 1.0 : 0.0);
                if (nw != act_[q]) { act_[q] = nw; bnd_[q] = nw < 0.0 ? lo : hi; sig_[q / M] = -1.0; chg = max(chg, q / M); }
            }
            chg = wmax_i(chg);
            rsync();
            t_dirty = max(t_dirty, chg);
            for (int it2 = 0; it2 < a.max_iter && !conv; ++it2) {
                ++iters;
                backward_sweep(t_dirty, t0);
                t_dirty = t0 - 1;
                policy_rollout(t0, us_, T);
                // largest feasible step along d = us - u over the free components
                double best = INF;
                int bq = 0x7fffffff;
                for (int q = t0 * M + lane; q < T * M; q += 64) {
                    if (act_[q] == 0.0) {
                        const double u = uu_[q], d = us_[q] - u;
                        double room = INF;
                        if (d > 0.0) room = (hi_[q] - u) / d;
                        else if (d < 0.0) room = (lo_[q] - u) / d;
                        if (room < best) { best = room; bq = q; }
                    }
                }
                const double alpha = wmin_d(best);
                if (alpha < 1.0) {
                    const int qb = wmin_i(best == alpha ? bq : 0x7fffffff);
                    for (int q = t0 * M + lane; q < T * M; q += 64) {
                        const double u = uu_[q], d = us_[q] - u;
                        if (q == qb) {
                            act_[q] = d > 0.0 ? 1.0 : -1.0;
                            uu_[q] = d > 0.0 ? hi_[q] : lo_[q];
                            bnd_[q] = uu_[q];
                            sig_[q / M] = -1.0;
                        } else {
                            uu_[q] = u + alpha * d;
                        }
                    }
                    rsync();
                    t_dirty = qb / M;
                    continue;
                }
                // full step: u = us; optimal if every pinned multiplier has the right sign
                double worst = 0.0;
                int wq = 0x7fffffff;
                for (int q = t0 * M + lane; q < T * M; q += 64) {
                    uu_[q] = us_[q];
                    const double ac = act_[q], mu = mu_[q];
                    const double viol = ac < 0.0 ? -mu : (ac > 0.0 ? mu : 0.0);
                    if (viol > worst) { worst = viol; wq = q; }
                }
                const double wmax = wmax_d(worst);
                if (wmax <= tol) {
                    rsync();
                    conv = true;
                } else {
                    const int qw = wmin_i(worst == wmax ? wq : 0x7fffffff);
                    if (lane == 0) { act_[qw] = 0.0; sig_[qw / M] = -1.0; }
                    rsync();
                    t_dirty = qw / M;
                }
            }
        }
#ifdef IRS_CBM_STAMPS
        st_acc[17] += iters;
#endif
        it_max = max(it_max, iters);
        n_fail += conv ? 0 : 1;
        full = !conv;
        if (tau == 0 && a.act_io != nullptr) {
            for (int q = lane; q < T * M; q += 64) a.act_io[q] = act_[q];
        }
        // first control of the tail solution (clipped) -> the plant
        if (lane < M) {
            const size_t q = (size_t)tau * M + lane;
            uctl[lane] = fmin(fmax(uu_[q], lo_[q]), hi_[q]);
        }
        wg_barrier();                                  // B(tau)
    }
    if (lane == 0) {
        a.info[0] = bad; a.info[1] = it_max; a.info[2] = n_fail;
    }
#ifdef IRS_CBM_STAMPS
    if (lane == 0 && stamps != nullptr) {
        st_acc[6] = __builtin_amdgcn_s_memtime() - st_begin;
        for (int k = 0; k < 24; ++k) stamps[k] = st_acc[k];
    }
#endif
}

constexpr size_t kLdsMax = 160 * 1024 - 512;

// diagnostic build: a device buffer for the stamps, printed (and reset) by irs_cbm_print_stamps()
#ifdef IRS_CBM_STAMPS
static long long* g_stamps = nullptr;
static long long* cbm_stamps() {
    if (g_stamps == nullptr) {
        (void)hipMalloc(reinterpret_cast<void**>(&g_stamps), 24 * sizeof(long long));
        (void)hipMemset(g_stamps, 0, 24 * sizeof(long long));
    }
    return g_stamps;
}
#else
static long long* cbm_stamps() { return nullptr; }
#endif

template <class Model, int KIND>
int launch_ctrlbox_mfma(const BoxArgs& a, double* ws, size_t ws_bytes, hipStream_t st) {
    using L = MfLayout<Model::NX, Model::NU>;
    const size_t rec = L::rec_doubles(a.T) * sizeof(double), small = L::small * sizeof(double);
    const bool in_lds = rec + small <= kLdsMax;
    if (!in_lds && (ws == nullptr || ws_bytes < rec)) {
        irs_set_error("irs_quasistatic_box_descent: horizon T=%d needs a %zu-byte workspace for the matrix-core "
                      "active-set solver (records do not fit LDS)", a.T, rec);
        return IRS_ERR_UNSUPPORTED;
    }
    const size_t bytes = in_lds ? rec + small : small;
    if (in_lds) {
        auto kern = ctrlbox_mfma_kernel<Model, KIND, true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            irs_set_error("irs_quasistatic_box_descent: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return IRS_ERR_HIP;
        }
        hipLaunchKernelGGL(kern, dim3(1), dim3(128), bytes, st, a, (double*)nullptr, cbm_stamps());
    } else {
        auto kern = ctrlbox_mfma_kernel<Model, KIND, false>;
        hipLaunchKernelGGL(kern, dim3(1), dim3(128), bytes, st, a, ws, cbm_stamps());
    }
    return IRS_OK;
}

}  // namespace

// 0 = the model does not fit one tile (not position controlled, or NHP + M > 16)
size_t irs_ctrlbox_mfma_record_bytes(int model, int T) {
    size_t r = 0;
    IRS_DISPATCH_MODEL(model, {
        if constexpr (has_u_into_x<Model>::value) {
            using L = MfLayout<Model::NX, Model::NU>;
            if constexpr (L::FITS) r = L::rec_doubles(T) * sizeof(double);
        }
    });
    return r;
}

// bytes of LDS the records need to stay on chip (0 = model unsupported); > kLdsMax: a workspace is needed
size_t irs_ctrlbox_mfma_lds_bytes(int model, int T) {
    size_t r = 0;
    IRS_DISPATCH_MODEL(model, {
        if constexpr (has_u_into_x<Model>::value) {
            using L = MfLayout<Model::NX, Model::NU>;
            if constexpr (L::FITS) r = (L::rec_doubles(T) + L::small) * sizeof(double);
        }
    });
    return r;
}

int irs_ctrlbox_mfma_launch(int model, const BoxArgs& a, int kind, double* ws, size_t ws_bytes, hipStream_t st) {
    int rc = IRS_ERR_UNSUPPORTED;
    IRS_DISPATCH_MODEL(model, {
        if constexpr (has_u_into_x<Model>::value) {
            if constexpr (MfLayout<Model::NX, Model::NU>::FITS) {
                rc = kind == KIND_ABS_M ? launch_ctrlbox_mfma<Model, KIND_ABS_M>(a, ws, ws_bytes, st)
                                        : launch_ctrlbox_mfma<Model, KIND_REL_M>(a, ws, ws_bytes, st);
            } else {
                irs_set_error("irs_quasistatic_box_descent: model %d does not fit the 16 x 16 tile", model);
            }
        } else {
            irs_set_error("irs_quasistatic_box_descent: model %d is not position controlled", model);
        }
    });
    return rc;
}

#ifdef IRS_CBM_STAMPS
// diagnostic build only: cycle totals of the LAST launch's solver wave
extern "C" void irs_cbm_print_stamps(void) {
    long long h[24];
    (void)hipDeviceSynchronize();
    if (g_stamps == nullptr || hipMemcpy(h, g_stamps, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
    fprintf(stderr, "[cbm stamps] backward %lld cyc / %lld steps; forward %lld cyc / %lld steps; wait for plant %lld cyc / "
                    "%lld tails; MPC loop %lld cyc; of backward: waiting for the prefetched step data %lld cyc\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    fprintf(stderr, "[cbm stamps] backward step phases: head+6 MFMA %lld, gather %lld, inverse+gains %lld, 3 MFMA+stores %lld cyc\n",
            h[8], h[9], h[10], h[11]);
    fprintf(stderr, "[cbm stamps] forward: %lld rollouts, %lld cyc inside their step loops; backward steps that re-swept a lazily skipped "
                    "pinned head: %lld (%lld of those sweeps released a component)\n", h[12], h[13], h[14], h[15]);
    fprintf(stderr, "[cbm stamps] active-set iterations over all tails: %lld; tails that left the primal-dual phase unconverged: %lld "
                    "(after %lld iterations)\n", h[17], h[16], h[18]);
    fprintf(stderr, "[cbm stamps] clipped rollouts: %lld plain steps in %lld chunks, %lld clipping steps\n", h[20], h[21], h[19]);
}
#endif
