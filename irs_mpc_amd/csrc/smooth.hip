// Randomised-smoothing linearisation: get_TV_matrices of
//   irs_lqr/irs_lqr_zero_order.py:38-63   (ZERO_ORDER_AB)
//   irs_lqr/irs_lqr_first_order.py:28-54  (FIRST_ORDER)
//   irs_lqr/quasistatic_dynamics.py:242-266 (ZERO_ORDER_B, u-only noise)
//   irs_lqr/irs_lqr_exact.py:15-31        (exact)
// as ONE launch: a streaming map-reduce over the (T x N) grid of independent one-step
// samples whose last-arriving workgroup per timestep finishes the job.
//
//   smooth_kernel   grid (nblk, T) x 256 threads.
//     1. every lane streams its samples' z=[dx|du] (f32, vector loads, consecutive
//        lanes read consecutive records), evaluates the model functor in f32 and
//        accumulates the P sufficient statistics of its timestep in registers;
//     2. the workgroup transpose-reduces them with wave shuffles + one LDS hop and
//        publishes its P partial sums (write-through stores), then takes a ticket on
//        the timestep's arrival counter;
//     3. the workgroup that draws the last ticket of timestep t re-reads the nblk
//        partials in a FIXED order (f64) -> sums[t][P]   (deterministic: no float
//        atomics, the arrival order never changes the summation order);
//     4. (single-GPU path) its first wave solves the Jacobi-scaled normal equations
//        by Cholesky in f64 -> A_t, B_t and c_t = f(x_t,u_t) - A_t x_t - B_t u_t.
//   With several GPUs step 4 is a separate launch (smooth_finalize_kernel) after the
//   all-reduce of `sums`.
//
// Inter-workgroup hand-off follows cdna_hip_programming.md Guideline 16: partials are
// stored sc1 (agent-scope relaxed atomic stores), every storing wave drains vmcnt, the
// workgroup barriers, one lane adds to the counter; the consumer does one agent-scope
// acquire + vmcnt(0) + barrier before any load of the partials.
#include "smooth_common.hpp"

namespace {

template <class Model, int MODE>
constexpr bool defer_samples() {
    return irs_contact_exact<Model>::value && MODE != IRS_SMOOTH_ZERO_ORDER_AB && contact_rows<Model>() <= 8;
}
// the f64 nominal step of workgroup 0 costs its wave about this many sample trips (parked-sample kernels)
constexpr int kNominalTrips = 2;

constexpr int kDeferRing = 128;     // entries per wave: < 64 waiting + <= 64 new ones

template <class Model, int MODE, bool RNG, bool FUSE, int BLOCK>
__global__ __launch_bounds__(BLOCK, (smooth_min_waves<Model, MODE>())) void smooth_kernel(SmoothArgs a) {
    using TR = SmoothTraits<Model, MODE>;
    constexpr int n = TR::n, m = TR::m, d = TR::d, NZ = TR::NZ, Z0 = TR::Z0, P = TR::P;
    constexpr int NW = BLOCK / 64;
    // matrix-core Gram path: zero-order, one 16-wide tile, too many statistics for registers
    constexpr bool USE_MFMA = MODE == IRS_SMOOTH_ZERO_ORDER_AB && d <= 16 && n <= 16 && TR::PP > 64 &&
                              BLOCK == kBlock;
    // exact contact models in the lane path: unfinished samples wait in a per-wave ring (see the DEFER branch)
    // Used by the u-only modes of the 8-row models (planar hand, T = 50: zero-order-B 102 -> 79 us at N = 1e4 and
    // 820 -> 606 us at 1e5; first-order 112 -> 76 us).  Measured and left on the plain path: the 12-row box models --
    // 39 % of their samples are unfinished after the first attempt, and W (144) + its factor live across the loop
    // spill 200 VGPRs: 123 -> 225 us.
    constexpr bool DEFER = defer_samples<Model, MODE>();
    // perturbed components of a sample in the u-only modes (NOT TR::NZ: that is the width of the least-squares design,
    // d for the first-order statistics) + the warm set
    constexpr int NPERT = d - Z0;
    constexpr int QE = NPERT + 1;
    __shared__ float defer_ring[DEFER ? NW * kDeferRing * QE : 1];
    __shared__ float red[NW * TR::PP];
    __shared__ float tile[USE_MFMA ? NW * 64 * 36 : 1];
    __shared__ double red64[TR::NGRP * P];
    __shared__ double tot[P];
    __shared__ FinalizeLds<Model, MODE> fin;
    __shared__ int s_ticket;

    const int t = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;

    // nominal point of this timestep (every lane keeps its own copy in registers)
    float xb[n], ub[m], f0[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xb[i] = (float)a.x_trj[(size_t)t * n + i];
#pragma unroll
    for (int j = 0; j < m; ++j) ub[j] = (float)a.u_trj[(size_t)t * m + j];
    // what the samples' f(x+dx,u+du) is measured from: the nominal step, or (TR::SUMZ) just xb
    if constexpr (MODE != IRS_SMOOTH_FIRST_ORDER && !TR::SUMZ) Model::template step<float>(a.p, xb, ub, f0);
    else {
#pragma unroll
        for (int i = 0; i < n; ++i) f0[i] = xb[i];
    }

    // workgroup 0 owns [0, chunk0), workgroup b >= 1 owns chunk0 + [(b-1) chunk, b chunk)
    const int s_begin = blk == 0 ? 0 : a.chunk0 + (blk - 1) * a.chunk;
    const int s_end = min(a.N, blk == 0 ? a.chunk0 : a.chunk0 + blk * a.chunk);
    constexpr bool NB = nominal_in_wg0<Model, MODE>();
    if constexpr (USE_MFMA) {
        // ---- matrix-core Gram accumulation (zero-order, d <= 16, many statistics) -------
        // The P = d(d+1)/2 + d n statistics are the products Z'Z and Z'dF over the sample
        // axis: exactly what v_mfma_f32_16x16x4_f32 contracts (k = 4 samples per issue),
        // with the accumulators in 8 registers per WAVE instead of P per LANE.  Each lane
        // evaluates one sample, stages [z | df] as a row of its wave's LDS tile (row stride
        // 36 dwords: conflict-free 16-byte writes), and the wave re-reads the tile in
        // operand layout (lane (i = l&15, k = l>>4) <- row 4 kk + k, column i); Z serves as
        // both the A and the B operand of Z'Z.  f32 MFMA is an exact k-ordered fmaf chain.
        typedef float v4f __attribute__((ext_vector_type(4)));
        constexpr int TS = 36;
        const int lane = tid & 63, wave = tid >> 6, col = lane & 15, rg = lane >> 4;
        float* my = tile + wave * 64 * TS;
        v4f aG = {0.f, 0.f, 0.f, 0.f}, aH = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = s_begin + wave * 64; s0 < s_end; s0 += BLOCK) {
            const int s = s0 + lane;
            const bool valid = s < s_end;
            float z[16], dfp[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { z[i] = 0.f; dfp[i] = 0.f; }
            if constexpr (RNG) {
                const unsigned long long gidx = a.sample_offset + (unsigned long long)s;
#pragma unroll
                for (int j = 0; j < (d + 3) / 4; ++j) {
                    float g[4];
                    philox_normal4(gidx, (unsigned)t, (unsigned)j, a.iter, a.seed, g);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (4 * j + c < d) z[4 * j + c] = __fmul_rn(g[c], a.std[4 * j + c]);   // rounded like a supplied f32 sample (no fma with the nominal point)
                }
            } else {
                const size_t row = (size_t)t * a.N + (valid ? s : s_end - 1);
                load_row<n>(a.dx + row * n, z);
                load_row<m>(a.du + row * m, z + n);
            }
#pragma unroll
            for (int i = 0; i < d; ++i) z[i] = valid ? z[i] : 0.f;
            float xs[n], us[m], fx[n];
#pragma unroll
            for (int i = 0; i < n; ++i) xs[i] = xb[i] + z[i];
#pragma unroll
            for (int j = 0; j < m; ++j) us[j] = ub[j] + z[n + j];
            Model::template step<float>(a.p, xs, us, fx);
#pragma unroll
            for (int k = 0; k < n; ++k) dfp[k] = fx[k] - f0[k];      // (!SUMZ: 0 for a zeroed slot)
            if constexpr (TR::SUMZ) {
                static_assert(!TR::SUMZ || n < 16, "column n of the dF tile carries the ones that sum z");
                dfp[n < 16 ? n : 0] = 1.f;                           // z of an invalid slot is 0
            }
            float4* rowp = reinterpret_cast<float4*>(my + lane * TS);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rowp[q] = make_float4(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]);
                rowp[4 + q] = make_float4(dfp[4 * q], dfp[4 * q + 1], dfp[4 * q + 2], dfp[4 * q + 3]);
            }
            wave_sync();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float av = my[(4 * kk + rg) * TS + col];
                const float bv = my[(4 * kk + rg) * TS + 16 + col];
                aG = __builtin_amdgcn_mfma_f32_16x16x4f32(av, av, aG, 0, 0, 0);
                aH = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, aH, 0, 0, 0);
            }
            wave_sync();
        }
        // accumulator (row = 4 (l>>4) + reg, col = l&15) -> this wave's row of `red`, P-order
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * rg + r, j = col;
            if (i < d && j < d && i <= j) red[wave * TR::PP + i * d - i * (i - 1) / 2 + (j - i)] = aG[r];
            if (i < d && j < n) red[wave * TR::PP + TR::NG + i * n + j] = aH[r];
            if constexpr (TR::SUMZ) {
                if (i < d && j == n) red[wave * TR::PP + TR::NH + i] = aH[r];
            }
        }
    } else if constexpr (DEFER) {
        // ---- exact contact models: finished samples accumulate at once, unfinished ones are parked ----------
        // irs_contact_step_try (contact_models.hpp) settles ~90 % of the samples with straight-line code; the
        // rest -- those whose guessed active set needs repair rounds or active-set steps, loops that run as long
        // as a wave's slowest lane -- are parked in this wave's LDS ring (the perturbation + the warm set, m + 1
        // dwords) and finished 64 at a time with every lane busy: as soon as 64 are waiting, and once more after
        // the wave's last sample.  Everything is wave-private and in lane order (ballot prefix, wave-uniform
        // head/tail): no barrier, no atomic, and the same sample lands in the same lane's accumulator in every
        // run -- the fixed-order sums stay bit-reproducible.
        float acc[TR::PP];
#pragma unroll
        for (int i = 0; i < TR::PP; ++i) acc[i] = 0.f;
        const int lane = tid & 63, wave = tid >> 6;
        float* ring = defer_ring + wave * (kDeferRing * QE);
        int qhead = 0, qtail = 0;                           // wave-uniform
        auto accumulate = [&](const float* z, const float* fx, const float* Bs, bool on) {
            if constexpr (TR::FIRST_B) {
#pragma unroll
                for (int q = 0; q < n * m; ++q) acc[q] += on ? Bs[q] : 0.f;
                // the derivative of a step whose command is NaN / Inf is finite (no row tests active): mark it
                bool nf = false;
#pragma unroll
                for (int j = 0; j < m; ++j) nf = nf || irs_nonfinite_bits(z[n + j]);
                if (on && nf) acc[0] = irs_poison();
            } else {
                float zz[NZ], df[n];
#pragma unroll
                for (int i = 0; i < NZ; ++i) zz[i] = on ? z[Z0 + i] : 0.f;
#pragma unroll
                for (int k = 0; k < n; ++k) df[k] = fx[k] - f0[k];
                int q = 0;
#pragma unroll
                for (int i = 0; i < NZ; ++i)
#pragma unroll
                    for (int j = i; j < NZ; ++j) { acc[q] = fmaf(zz[i], zz[j], acc[q]); ++q; }
#pragma unroll
                for (int i = 0; i < NZ; ++i)
#pragma unroll
                    for (int k = 0; k < n; ++k) { acc[q] = fmaf(zz[i], df[k], acc[q]); ++q; }
                if constexpr (TR::SUMZ) {
#pragma unroll
                    for (int i = 0; i < NZ; ++i) { acc[q] += zz[i]; ++q; }
                }
            }
        };
        // One loop, two kinds of trips (the choice is wave-uniform): a FRESH trip takes the wave's next 64 samples
        // through the first attempt; a FLUSH trip takes up to 64 parked samples through the full method.  Both share
        // the assembly of the step QP in front and the primal recovery / derivative / accumulation behind, so the
        // kernel holds one copy of each (two complete inlined steps cost the first-order kernel 137 spilled VGPRs).
        constexpr int NC = Model::NC;
        // trip k of wave w takes 64-sample block 4 k + w of the workgroup; in workgroup 0 the last wave sits out
        // after a.wg0_rr trips (it evaluates the f64 nominal step instead) and the other three share the rest
        int kt = 0;                                         // wave-uniform
        const int rr = blk == 0 ? a.wg0_rr : 0x7fffffff;
        auto block_of = [&](int k) {
            if (k < rr) return NW * k + wave;
            return wave == NW - 1 ? 0x3fffffff : NW * rr + (NW - 1) * (k - rr) + wave;
        };
        int sb = s_begin + 64 * block_of(0);
        constexpr bool HOIST = !TR::FIRST_B;
        float q[n], b0[n], J[NC][n], phi[NC], Dinv[n];
        if constexpr (HOIST) Model::template assemble<float>(a.p, xb, ub, q, Dinv, b0, J, phi);
        while (true) {
            const bool fresh = sb < s_end;
            const int pending = qtail - qhead;
            const bool flush = pending >= 64 || (!fresh && pending > 0);
            if (!fresh && !flush) break;
            float z[d];
            bool on;
            unsigned wm = 0u;
            int take = 0;
            if (flush) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                take = min(64, pending);
                on = lane < take;
                const int slot = (qhead + (on ? lane : 0)) & (kDeferRing - 1);
#pragma unroll
                for (int i = 0; i < Z0; ++i) z[i] = 0.f;
#pragma unroll
                for (int i = 0; i < NPERT; ++i) z[Z0 + i] = ring[slot * QE + i];
                wm = __float_as_uint(ring[slot * QE + NPERT]);
            } else {
                const int sidx = sb + lane;
                on = sidx < s_end;
                if constexpr (RNG) {
                    constexpr int j0 = Z0 / 4;
                    const unsigned long long gidx = a.sample_offset + (unsigned long long)sidx;
#pragma unroll
                    for (int j = j0; j < (d + 3) / 4; ++j) {
                        float g[4];
                        philox_normal4(gidx, (unsigned)t, (unsigned)j, a.iter, a.seed, g);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (4 * j + c < d) z[4 * j + c] = __fmul_rn(g[c], a.std[4 * j + c]);   // rounded like a supplied f32 sample (no fma with the nominal point)
                    }
#pragma unroll
                    for (int i = 0; i < Z0; ++i) z[i] = 0.f;
                } else {
                    static_assert(Z0 == n && NPERT == m, "the parked-sample loop serves the u-only modes");
#pragma unroll
                    for (int i = 0; i < n; ++i) z[i] = 0.f;
                    // (a register prefetch of the next trip's sample and staging 8 trips through LDS were both measured:
                    // no change -- the load at the head of a trip is not what a trip waits for)
                    const size_t row = (size_t)t * a.N + (on ? sidx : s_end - 1);
                    load_row<NPERT>(a.du + row * NPERT, z + n);
                }
            }
            float xs[n], us[m], fx[n], Bs[TR::FIRST_B ? n * m : 1];
#pragma unroll
            for (int i = 0; i < n; ++i) xs[i] = xb[i] + z[i];
#pragma unroll
            for (int j = 0; j < m; ++j) us[j] = ub[j] + z[n + j];
            // the state is not perturbed in these modes: geometry (q, D, J, phi) assembled ONCE, before the loop; per
            // sample only the actuated entries of b -- the same expression the model's assemble evaluates
            // (zero-order-B: the compiler left ~1 900 cycles of assembly in every trip -- 69.6 -> 63.1 us; in the
            // first-order kernel it hoists by itself and the explicit form costs registers: 68.7 -> 71.3 us, so there
            // the model's assemble stays in the loop)
            float qn[n], bq[n], W[NC][NC], lam[NC];
            if constexpr (HOIST) {
#pragma unroll
                for (int k = 0; k < n; ++k) bq[k] = b0[k];
#pragma unroll
                for (int j = 0; j < m; ++j) bq[Model::act(j)] = Model::template stiffness<float>(a.p, j) * (q[Model::act(j)] - us[j]);
            } else {
                Model::template assemble<float>(a.p, xs, us, q, Dinv, bq, J, phi);
            }
            bool fin_ = true;
            if (flush) {
                irs_contact_qp_dual_exact<float, n, NC>(Dinv, bq, J, phi, W, lam, &wm);
                qhead += take;
            } else {
                unsigned mask = 0u;
                fin_ = irs_contact_qp_dual_exact_try<float, n, NC>(Dinv, bq, J, phi, W, lam, &mask);
                const bool hard = on && !fin_;
                const unsigned long long bal = __ballot(hard);
                if (hard) {
                    const int slot = (qtail + __popcll(bal & ((1ull << lane) - 1ull))) & (kDeferRing - 1);
#pragma unroll
                    for (int i = 0; i < NPERT; ++i) ring[slot * QE + i] = z[Z0 + i];
                    ring[slot * QE + NPERT] = __uint_as_float(mask);
                }
                qtail += __popcll(bal);
                ++kt;
                const int nb_ = block_of(kt);
                sb = nb_ >= 0x3fffffff ? s_end : s_begin + 64 * nb_;
            }
            irs_contact_qp_primal<float, n, NC>(q, Dinv, bq, J, lam, qn);
#pragma unroll
            for (int k = 0; k < n; ++k) fx[Model::perm(k)] = qn[k];
            if constexpr (TR::FIRST_B) {
                int actc[m];
#pragma unroll
                for (int j = 0; j < m; ++j) actc[j] = Model::act(j);
                float Bint[n][m], Aint[1][n];
                irs_contact_qp_grad<float, n, NC, m, false>(Dinv, J, W, lam, actc, Bint, Aint);
#pragma unroll
                for (int k = 0; k < n; ++k)
#pragma unroll
                    for (int j = 0; j < m; ++j) Bs[Model::perm(k) * m + j] = Bint[k][j];
            }
            accumulate(z, fx, Bs, on && fin_);
        }
        block_reduce_lds<P, NW>(acc, red);
    } else {
        float acc[TR::PP];
#pragma unroll
        for (int i = 0; i < TR::PP; ++i) acc[i] = 0.f;
        constexpr int NJC = compact_jac_len<Model>();      // 0: the model has no hand-derived compact Jacobian
        float accj[irs_reduce_pad(NJC + 1)];                // + the number of samples this lane summed
#pragma unroll
        for (int i = 0; i < irs_reduce_pad(NJC + 1); ++i) accj[i] = 0.f;

        // Sample loop.  U samples per lane are loaded together (independent loads in
        // flight), then evaluated; out-of-range slots are clamped to a valid address and
        // zeroed (a zero perturbation contributes exactly nothing to the zero-order sums).
        constexpr int U = (TR::LIGHT && !RNG) ? 4 : 1;
        // kernels whose workgroup 0 evaluates the f64 nominal step deal 64-sample blocks to WAVES (block_of, as in
        // the parked-sample loop above): trip k of wave w takes block NW k + w; in workgroup 0 the last wave sits out
        // after a.wg0_rr trips.  Everything else: the plain strided loop.
        const int rr_ = (NB && blk == 0) ? a.wg0_rr : 0x7fffffff;
        auto next_s0 = [&](int k) {
            if constexpr (NB) {
                if (k < rr_) return s_begin + tid + k * BLOCK;
                if ((tid >> 6) == NW - 1) return s_end;
                return s_begin + 64 * (NW * rr_ + (NW - 1) * (k - rr_)) + tid;
            } else {
                return s_begin + tid + k * BLOCK;
            }
        };
        // One sample per trip and supplied samples (the 16-dimensional analytic models): the NEXT trip's row is requested
        // before this trip's arithmetic starts (quadrotor first-order, N = 1e5: 93 -> 85 us.  Two rows ahead: 92 us --
        // the 16 extra registers and moves cost more than the second request in flight gains)
        constexpr bool PREF = !RNG && !NB && U == 1;
        float zpre[PREF ? d : 1];
        auto request_row = [&](int s) {
            if constexpr (PREF) {
                const size_t row = (size_t)t * a.N + (s < s_end ? s : s_end - 1);
                if constexpr (Z0 == 0) load_row<n>(a.dx + row * n, zpre);
                else {
#pragma unroll
                    for (int i = 0; i < n; ++i) zpre[i] = 0.f;
                }
                load_row<m>(a.du + row * m, zpre + n);
            }
        };
        if (s_begin < s_end) request_row(s_begin + tid);
        int kt_ = 0;
        for (int s0 = s_begin + tid; s0 < s_end; s0 = (NB && U == 1) ? next_s0(++kt_) : s0 + BLOCK * U) {
            float zz[U][d];
            bool valid[U];
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                const int s = s0 + uu * BLOCK;
                valid[uu] = s < s_end;
                if constexpr (RNG) {
                    constexpr int j0 = Z0 / 4;
                    const unsigned long long gidx = a.sample_offset + (unsigned long long)s;
#pragma unroll
                    for (int j = j0; j < (d + 3) / 4; ++j) {
                        float g[4];
                        philox_normal4(gidx, (unsigned)t, (unsigned)j, a.iter, a.seed, g);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (4 * j + c < d) zz[uu][4 * j + c] = __fmul_rn(g[c], a.std[4 * j + c]);   // rounded like a supplied f32 sample (no fma with the nominal point)
                    }
#pragma unroll
                    for (int i = 0; i < Z0; ++i) zz[uu][i] = 0.f;
                } else if constexpr (PREF) {
#pragma unroll
                    for (int i = 0; i < d; ++i) zz[uu][i] = zpre[i];
                    request_row(s0 + BLOCK);
                } else {
                    const size_t row = (size_t)t * a.N + (valid[uu] ? s : s_end - 1);
                    if constexpr (Z0 == 0) load_row<n>(a.dx + row * n, zz[uu]);
                    else {
#pragma unroll
                        for (int i = 0; i < n; ++i) zz[uu][i] = 0.f;
                    }
                    load_row<m>(a.du + row * m, zz[uu] + n);
                }
            }
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                float z[d];
#pragma unroll
                for (int i = 0; i < d; ++i) z[i] = (U == 1 || valid[uu]) ? zz[uu][i] : 0.f;
                float xs[n], us[m], fx[n];
#pragma unroll
                for (int i = 0; i < n; ++i) xs[i] = xb[i] + z[i];
#pragma unroll
                for (int j = 0; j < m; ++j) us[j] = ub[j] + z[n + j];

                if constexpr (TR::FIRST_B) {
                    float Bs[n * m];
                    irs_contact_step_grad<Model, float, false>(a.p, xs, us, fx, Bs, nullptr);
#pragma unroll
                    for (int q = 0; q < n * m; ++q) acc[q] += Bs[q];
                    bool nf = false;
#pragma unroll
                    for (int j = 0; j < m; ++j) nf = nf || irs_nonfinite_bits(z[n + j]);
                    if (nf) acc[0] = irs_poison();
                } else if constexpr (MODE == IRS_SMOOTH_FIRST_ORDER && has_compact_jac<Model>::value) {
                    // hand-derived Jacobian, only its sample-dependent entries (models.hpp, step_jac): the full n x d
                    // sum is put together once per workgroup after the loop
                    float Jc[NJC];
                    Model::template step_jac<float>(a.p, xs, us, fx, Jc);
                    const float w = (U == 1 || valid[uu]) ? 1.f : 0.f;
#pragma unroll
                    for (int q = 0; q < NJC; ++q) accj[q] = fmaf(w, Jc[q], accj[q]);
                    accj[NJC] += w;
                } else if constexpr (MODE == IRS_SMOOTH_FIRST_ORDER) {
                    float J[n * d];
                    model_jacobian<Model, float>(a.p, xs, us, fx, J);
                    const float w = (U == 1 || valid[uu]) ? 1.f : 0.f;
#pragma unroll
                    for (int q = 0; q < n * d; ++q) acc[q] = fmaf(w, J[q], acc[q]);
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
                    if constexpr (TR::SUMZ) {
#pragma unroll
                        for (int i = 0; i < NZ; ++i) { acc[q] += z[Z0 + i]; ++q; }
                    }
                }
            }
        }

        if constexpr (MODE == IRS_SMOOTH_FIRST_ORDER && has_compact_jac<Model>::value) {
            // the workgroup reduces the COMPACT sums (34 numbers, not 192), adds the waves' rows in a fixed order, and
            // only then lays the n x d sum out as `red`'s row 0 (the other rows zero)
            constexpr int PC = irs_reduce_pad(NJC + 1);
            __shared__ float redc[NW * PC + PC];
            block_reduce_lds<NJC + 1, NW>(accj, redc);
            __syncthreads();
            if (tid <= NJC) {
                float tsum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tsum += redc[w * PC + tid];
                redc[NW * PC + tid] = tsum;
            }
            __syncthreads();
            for (int q = tid; q < NW * TR::PP; q += BLOCK) {
                const int w = q / TR::PP, e = q - w * TR::PP;
                red[q] = (w == 0 && e < P) ? Model::template expand_entry<float>(a.p, redc + NW * PC, redc[NW * PC + NJC], e) : 0.f;
            }
        } else {
            // ---- workgroup reduction: registers -> shuffles -> LDS ---------------------
            block_reduce_lds<P, NW>(acc, red);
        }
    }
    if constexpr (NB) {
        // (a lone fused workgroup lets its solve evaluate the step itself: same cost, no round trip)
        constexpr int NOMW = NW - 1;                        // the wave that evaluates it: the one that is dealt fewer
                                                           // sample trips (a.wg0_rr)
        if (blk == 0 && (a.nblk > 1 || !FUSE) && (tid >> 6) == NOMW) {
            double x64[n], u64[m], f64[n];
#pragma unroll
            for (int i = 0; i < n; ++i) x64[i] = a.x_trj[(size_t)t * n + i];
#pragma unroll
            for (int j = 0; j < m; ++j) u64[j] = a.u_trj[(size_t)t * m + j];
            Model::template step<double>(a.p, x64, u64, f64);
            if ((tid & 63) == 0) {
#pragma unroll
                for (int i = 0; i < n; ++i)
                    __hip_atomic_store(a.fnom + (size_t)t * n + i, f64[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    smooth_finish<Model, MODE, FUSE, BLOCK>(a, red, red64, tot, fin, s_ticket, t, blk, tid,
                                            (NB && a.nblk > 1) ? a.fnom + (size_t)t * n : nullptr);
}

// Stand-alone solve (after an all-reduce of `sums`): one wave per timestep.
template <class Model, int MODE>
__global__ __launch_bounds__(64) void smooth_finalize_kernel(SmoothArgs a) {
    __shared__ FinalizeLds<Model, MODE> fin;
    const int t = blockIdx.x;
    // a.fnom (optional): the f64 nominal steps a preceding irs_smooth_accumulate left in its workspace
    const double* fnom = (nominal_in_wg0<Model, MODE>() && a.fnom) ? a.fnom + (size_t)t * SmoothTraits<Model, MODE>::n
                                                                    : nullptr;
    finalize_timestep<Model, MODE>(a.p, a.x_trj, a.u_trj, a.sums + (size_t)t * SmoothTraits<Model, MODE>::P,
                                   a.n_total, t, threadIdx.x, fin, a.At, a.Bt, a.ct, a.info, fnom);
}

template <class Model>
__global__ __launch_bounds__(64) void exact_linearize_kernel(ModelParams p, const double* x_trj,
                                                             const double* u_trj, double* At, double* Bt,
                                                             double* ct, int T) {
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

// Planner knobs (environment overrides are for tuning experiments only).
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    int v = e ? atoi(e) : dflt;
    return v < 1 ? 1 : v;
}
int tune_spt() { static int v = env_int("IRS_SPT", 32); return v; }                 // samples per lane aimed for
int tune_single_max() { static int v = env_int("IRS_SINGLE_MAX", 16384); return v; } // <= : one WG per t
int tune_max_wg() { static int v = env_int("IRS_MAX_WG", 1024); return v; }         // grid cap, light kernels
int tune_max_wg_heavy() { static int v = env_int("IRS_MAX_WG_HEAVY", 256); return v; }  // 1 wave/SIMD kernels
int tune_min_wg() { static int v = env_int("IRS_MIN_WG", 256); return v; }          // fill the CUs

constexpr int kBigBlock = 1024;

// Chooses the launch geometry for (T, N).  `light` = the kernel's accumulators fit a
// 1024-thread workgroup (SmoothTraits::LIGHT).  Measured on MI355X (profiles/): every extra
// workgroup costs more (its hand-off: write-through stores + ticket) than it gains in
// streaming parallelism once the CUs are covered, so grids stay SMALL:
//  * light, samples supplied, N <= tune_single_max(): ONE 1024-thread workgroup per
//    timestep -- no inter-workgroup hand-off at all;
//  * otherwise 256-thread workgroups, ~tune_spt() samples per lane, at least enough
//    workgroups to cover the CUs (while each lane still has >= 4 samples) and at most
//    ~4 per CU (light kernels) or 1 per CU (kernels that hold 1 wave per SIMD).
void plan_grid(int T, int N, bool light, bool rng, int* chunk, int* nblk, int* block, bool contact = false) {
    if (light && !rng && N <= tune_single_max()) {
        *block = kBigBlock;
        *nblk = 1;
        *chunk = (N + kBigBlock - 1) / kBigBlock * kBigBlock;
        return;
    }
    *block = kBlock;
    if (T < 1) T = 1;
    const int spt = tune_spt();
    int nb = (N + kBlock * spt - 1) / (kBlock * spt);
    int fill = (tune_min_wg() + T - 1) / T;               // blocks per t that cover the CUs
    // ... keeping >= 4 samples per lane -- 1 for a contact kernel, whose sample costs more than a workgroup's
    // hand-off (planar hand N = 1000: one workgroup per time step 51 us, four 2x faster)
    int by4 = contact ? (N + kBlock - 1) / kBlock : (N + kBlock * 4 - 1) / (kBlock * 4);
    if (fill > by4) fill = by4;
    if (nb < fill) nb = fill;
    // heavy kernels: 1 workgroup per CU, 2 once there is enough work to hide the tail.  Contact
    // kernels never: since the active-set polish they need more than 256 registers (VGPRs + AGPRs), one
    // wave per SIMD is all a CU can hold, and a second workgroup per CU would only queue behind the first
    // (before the polish, two per CU won 7 % from N = 1e5 on and lost below that: profiles/).
    const long long two_per_cu = contact ? (1LL << 62) : 2000000;
    const int heavy_cap = tune_max_wg_heavy() * ((long long)N * T >= two_per_cu ? 2 : 1);
    int max_blk = (light ? tune_max_wg() : heavy_cap) / T;
    if (max_blk < 1) max_blk = 1;
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
bool light_m(int mode) {
    switch (mode) {
        case IRS_SMOOTH_ZERO_ORDER_AB: return SmoothTraits<Model, IRS_SMOOTH_ZERO_ORDER_AB>::LIGHT;
        case IRS_SMOOTH_FIRST_ORDER: return SmoothTraits<Model, IRS_SMOOTH_FIRST_ORDER>::LIGHT;
        case IRS_SMOOTH_ZERO_ORDER_B: return SmoothTraits<Model, IRS_SMOOTH_ZERO_ORDER_B>::LIGHT;
    }
    return false;
}

bool is_light(int model, int mode) {
    switch (model) {
        case IRS_MODEL_PENDULUM: return light_m<PendulumModel>(mode);
        case IRS_MODEL_QUADROTOR: return light_m<QuadrotorModel>(mode);
        case IRS_MODEL_BICYCLE: return light_m<BicycleModel>(mode);
        case IRS_MODEL_THREE_CART: return light_m<ThreeCartModel>(mode);
        case IRS_MODEL_PLANAR_HAND: return light_m<PlanarHandModel>(mode);
        case IRS_MODEL_BOX_PIVOT: return light_m<BoxPivotModel>(mode);
        case IRS_MODEL_BOX_ON_BOX: return light_m<BoxOnBoxModel>(mode);
        case IRS_MODEL_BOX_PUSH: return light_m<BoxPushModel>(mode);
        case IRS_MODEL_PLANAR_HAND_EXACT: return light_m<PlanarHandExactModel>(mode);
        case IRS_MODEL_BOX_PIVOT_EXACT: return light_m<BoxPivotExactModel>(mode);
        case IRS_MODEL_BOX_PUSH_EXACT: return light_m<BoxPushExactModel>(mode);
    }
    return false;
}

// contact models, in every smoothing mode (nominal_in_wg0<Model, MODE>)
bool has_nominal_in_wg0(int model, int /*mode*/) {
    bool r = false;
    IRS_DISPATCH_MODEL(model, { r = !Model::HAS_JACOBIAN; });
    return r;
}

// what the f64 nominal step costs the wave of workgroup 0 that evaluates it, in 64-sample trips of the same kernel
// (measured per kernel family; IRS_NOMINAL_TRIPS overrides for tuning)
int nominal_trips(int model, int mode);

// smooth_kernel's DEFER path (defer_samples<Model, MODE>)
bool uses_parked_samples(int model, int mode) {
    bool r = false;
    IRS_DISPATCH_MODEL(model, {
        r = mode == IRS_SMOOTH_FIRST_ORDER ? defer_samples<Model, IRS_SMOOTH_FIRST_ORDER>()
                                           : mode == IRS_SMOOTH_ZERO_ORDER_B ? defer_samples<Model, IRS_SMOOTH_ZERO_ORDER_B>() : false;
    });
    return r;
}

int nominal_trips(int model, int mode) {
    static int ov = env_int("IRS_NOMINAL_TRIPS", -1);
    if (ov >= 0) return ov;
    (void)model; (void)mode;
    return kNominalTrips;      // measured: 1-3 trips are equal for every contact kernel (planar hand exact / sweeps /
                               // first-order, box pivoting); 4 and more lose a trip
}

template <class Model>
int sums_len_m(int mode) {
    switch (mode) {
        case IRS_SMOOTH_ZERO_ORDER_AB: return sums_len_t<Model, IRS_SMOOTH_ZERO_ORDER_AB>();
        case IRS_SMOOTH_FIRST_ORDER: return sums_len_t<Model, IRS_SMOOTH_FIRST_ORDER>();
        case IRS_SMOOTH_ZERO_ORDER_B: return sums_len_t<Model, IRS_SMOOTH_ZERO_ORDER_B>();
    }
    return -1;
}

template <class Model, int MODE, int BLOCK>
void launch_smooth_b(const SmoothArgs& a, bool rng, bool fuse, hipStream_t st) {
    dim3 grid(a.nblk, a.T), block(BLOCK);
    if (rng) {
        if (fuse) hipLaunchKernelGGL((smooth_kernel<Model, MODE, true, true, BLOCK>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((smooth_kernel<Model, MODE, true, false, BLOCK>), grid, block, 0, st, a);
    } else {
        if (fuse) hipLaunchKernelGGL((smooth_kernel<Model, MODE, false, true, BLOCK>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((smooth_kernel<Model, MODE, false, false, BLOCK>), grid, block, 0, st, a);
    }
}

template <class Model, int MODE>
void launch_smooth_m(const SmoothArgs& a, bool rng, bool fuse, hipStream_t st) {
    if constexpr (SmoothTraits<Model, MODE>::LIGHT) {
        if (a.block == kBigBlock) { launch_smooth_b<Model, MODE, kBigBlock>(a, rng, fuse, st); return; }
    }
    launch_smooth_b<Model, MODE, kBlock>(a, rng, fuse, st);
}

template <class Model>
int launch_smooth(int mode, const SmoothArgs& a, bool rng, bool fuse, hipStream_t st) {
    switch (mode) {
        case IRS_SMOOTH_ZERO_ORDER_AB: launch_smooth_m<Model, IRS_SMOOTH_ZERO_ORDER_AB>(a, rng, fuse, st); break;
        case IRS_SMOOTH_FIRST_ORDER: launch_smooth_m<Model, IRS_SMOOTH_FIRST_ORDER>(a, rng, fuse, st); break;
        case IRS_SMOOTH_ZERO_ORDER_B: launch_smooth_m<Model, IRS_SMOOTH_ZERO_ORDER_B>(a, rng, fuse, st); break;
        default: return IRS_ERR_UNSUPPORTED;
    }
    return IRS_OK;
}

template <class Model>
int launch_finalize(int mode, const SmoothArgs& a, hipStream_t st) {
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

struct SmoothOut {   // finalize outputs; all null = accumulate only
    double* At; double* Bt; double* ct; int* info; long long n_total;
};

int smooth_common(int model, const double* params, int n_params, int mode, int T, int N,
                  const double* x_trj, const double* u_trj, SmoothArgs& a, bool rng, double* sums,
                  const SmoothOut* out, void* workspace, size_t workspace_bytes, void* stream) {
    IRS_CHECK_ARG(T > 0 && N > 0, "T and N must be positive");
    IRS_CHECK_ARG(x_trj && u_trj && sums && workspace, "null pointer");
    IRS_CHECK_ARG(mode >= 0 && mode <= 2, "unknown smoothing mode");
    IRS_CHECK_ARG((size_t)T * sizeof(int) <= kCounterBytes, "T too large for the arrival counters (max 1024)");
    int rc = irs_load_params(model, params, n_params, &a.p);
    if (rc != IRS_OK) return rc;
    size_t need = irs_smooth_workspace_bytes(model, mode, T, N);
    if (workspace_bytes < need) {
        irs_set_error("irs_smooth: workspace %zu < %zu bytes", workspace_bytes, need);
        return IRS_ERR_WORKSPACE;
    }
    a.x_trj = x_trj; a.u_trj = u_trj;
    a.T = T; a.N = N;
    if (irs_smooth_ug_supported(model, mode)) {
        // exact 8-row contact model, u-only mode: the uniform-geometry pass (smooth_ug.hip)
        a.nblk = irs_smooth_ug_nblk(T, N);
        a.block = 512;
        a.chunk = a.chunk0 = 0;
        a.wg0_rr = 0x7fffffff;
        a.diag = getenv("IRS_DIAG") ? atoi(getenv("IRS_DIAG")) : 0;      // timing experiments only
        a.counters = static_cast<int*>(workspace);
        a.fnom = reinterpret_cast<double*>(static_cast<char*>(workspace) + kCounterBytes);
        a.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + kCounterBytes + fnom_bytes(T));
        a.sums = sums;
        if (out != nullptr) {
            IRS_CHECK_ARG(out->At && out->Bt && out->ct && out->info && out->n_total > 0, "null output pointer");
            a.At = out->At; a.Bt = out->Bt; a.ct = out->ct; a.info = out->info;
            a.n_total = (double)out->n_total;
        }
        rc = irs_smooth_ug_launch(model, mode, a, rng, out != nullptr, static_cast<hipStream_t>(stream));
        if (rc != IRS_OK) return rc;
        IRS_CHECK_LAUNCH();
        return IRS_OK;
    }
    plan_grid(T, N, is_light(model, mode), rng, &a.chunk, &a.nblk, &a.block, has_nominal_in_wg0(model, mode));
    a.chunk0 = a.chunk;
    a.wg0_rr = 0x7fffffff;
    bool planned = false;
    if (a.nblk >= 2 && a.block == kBlock && has_nominal_in_wg0(model, mode)) {
        // contact kernels balance by WAVE trips (64 samples): B blocks plus the nominal step's kNominalTrips
        // over 4 nblk waves -> tt trips each; workgroup 0: tt - kNominalTrips rounds over its four waves, then
        // kNominalTrips rounds over three.  (Chunks rounded to whole workgroup trips, as below, left three of the
        // five workgroups of the benchmark's N = 1e4 with 9 trips and one with 5 + the nominal step.)
        const int NW = kBlock / 64, B = (N + 63) / 64, nw = a.nblk * NW;
        const int kNom = nominal_trips(model, mode);
        const int tt = (B + kNom + nw - 1) / nw, rr = tt - kNom;
        if (rr >= 1) {
            const int c0b = NW * rr + (NW - 1) * kNom;
            const int restb = B > c0b ? (B - c0b + (a.nblk - 1) - 1) / (a.nblk - 1) : 0;
            if (restb <= NW * tt) {
                a.chunk0 = c0b * 64;
                a.chunk = restb > 0 ? restb * 64 : 64;
                a.wg0_rr = rr;
                planned = true;
            }
        }
    }
    if (!planned && a.nblk >= 2 && a.block == kBlock && has_nominal_in_wg0(model, mode)) {
        // workgroup 0 gives up kNominalCost samples per lane and evaluates the f64 nominal step
        int c0 = a.chunk - kNominalCost * kBlock;
        if (c0 < kBlock) c0 = kBlock;
        int rest = (N - c0 + (a.nblk - 1) - 1) / (a.nblk - 1);
        rest = (rest + kBlock - 1) / kBlock * kBlock;
        if (c0 < a.chunk && c0 + (long long)(a.nblk - 1) * rest >= N && rest <= a.chunk + kBlock) {
            a.chunk0 = c0;
            a.chunk = rest;
        }
    }
    { static int dg = getenv("IRS_DIAG") ? atoi(getenv("IRS_DIAG")) : 0; a.diag = dg; }
    a.counters = static_cast<int*>(workspace);
    a.fnom = reinterpret_cast<double*>(static_cast<char*>(workspace) + kCounterBytes);
    a.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + kCounterBytes + fnom_bytes(T));
    a.sums = sums;
    const bool fuse = out != nullptr;
    if (fuse) {
        IRS_CHECK_ARG(out->At && out->Bt && out->ct && out->info && out->n_total > 0, "null output pointer");
        a.At = out->At; a.Bt = out->Bt; a.ct = out->ct; a.info = out->info;
        a.n_total = (double)out->n_total;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, { rc = launch_smooth<Model>(mode, a, rng, fuse, st); });
    if (rc != IRS_OK) return rc;
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

// modes that perturb u only never read dx / std_x
bool u_noise_only(int model, int mode) {
    if (mode == IRS_SMOOTH_ZERO_ORDER_B) return true;
    bool contact = false;
    IRS_DISPATCH_MODEL(model, { contact = !Model::HAS_JACOBIAN; });
    return contact && mode == IRS_SMOOTH_FIRST_ORDER;
}

int fill_rng(int model, int mode, const double* std_x, const double* std_u, uint64_t seed, uint32_t iter,
             uint64_t sample_offset, SmoothArgs& a) {
    IRS_CHECK_ARG(std_u != nullptr, "std_u is null");
    IRS_CHECK_ARG(std_x != nullptr || u_noise_only(model, mode), "std_x is null");
    int n, m, np;
    int rc = irs_model_info(model, &n, &m, &np);
    if (rc != IRS_OK) return rc;
    for (int i = 0; i < n; ++i) a.std[i] = std_x ? (float)std_x[i] : 0.f;
    for (int j = 0; j < m; ++j) a.std[n + j] = (float)std_u[j];
    a.seed = seed; a.iter = iter; a.sample_offset = sample_offset;
    return IRS_OK;
}

int check_samples(const float* dx, const float* du, int model, int mode) {
    IRS_CHECK_ARG(du != nullptr, "du is null");
    IRS_CHECK_ARG(dx != nullptr || u_noise_only(model, mode), "dx is null");
    IRS_CHECK_ARG((reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (reinterpret_cast<uintptr_t>(du) & 15) == 0,
                  "dx/du must be 16-byte aligned");
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
    int chunk, nblk, block;
    plan_grid(T, N, is_light(model, mode), /*rng=*/true, &chunk, &nblk, &block,       // the larger grid
              has_nominal_in_wg0(model, mode));
    if (irs_smooth_ug_supported(model, mode)) {
        const int ug = irs_smooth_ug_nblk(T, N);
        if (ug > nblk) nblk = ug;
    }
    return kCounterBytes + fnom_bytes(T) + (size_t)T * nblk * ((P + 3) / 4 * 4) * sizeof(float);
}

int irs_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    IRS_CHECK_ARG(workspace != nullptr && workspace_bytes >= (size_t)kCounterBytes, "workspace too small");
    hipError_t e = hipMemsetAsync(workspace, 0, kCounterBytes, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) { irs_set_error("irs_workspace_init: %s", hipGetErrorString(e)); return IRS_ERR_HIP; }
    return IRS_OK;
}

int irs_smooth_accumulate(int model, const double* params, int n_params, int mode, int T, int N,
                          const double* x_trj, const double* u_trj, const float* dx,
                          const float* du, double* sums, void* workspace,
                          size_t workspace_bytes, void* stream) {
    int rc = check_samples(dx, du, model, mode);
    if (rc != IRS_OK) return rc;
    SmoothArgs a;
    memset(&a, 0, sizeof(a));
    a.dx = dx; a.du = du;
    return smooth_common(model, params, n_params, mode, T, N, x_trj, u_trj, a, false, sums, nullptr,
                         workspace, workspace_bytes, stream);
}

int irs_smooth_accumulate_rng(int model, const double* params, int n_params, int mode, int T, int N,
                              const double* x_trj, const double* u_trj, const double* std_x,
                              const double* std_u, uint64_t seed, uint32_t iter,
                              uint64_t sample_offset, double* sums, void* workspace,
                              size_t workspace_bytes, void* stream) {
    SmoothArgs a;
    memset(&a, 0, sizeof(a));
    int rc = fill_rng(model, mode, std_x, std_u, seed, iter, sample_offset, a);
    if (rc != IRS_OK) return rc;
    return smooth_common(model, params, n_params, mode, T, N, x_trj, u_trj, a, true, sums, nullptr,
                         workspace, workspace_bytes, stream);
}

int irs_smooth(int model, const double* params, int n_params, int mode, int T, int N,
               const double* x_trj, const double* u_trj, const float* dx, const float* du,
               double* sums, double* At, double* Bt, double* ct, int* info, void* workspace,
               size_t workspace_bytes, void* stream) {
    int rc = check_samples(dx, du, model, mode);
    if (rc != IRS_OK) return rc;
    SmoothArgs a;
    memset(&a, 0, sizeof(a));
    a.dx = dx; a.du = du;
    SmoothOut out{At, Bt, ct, info, (long long)N};
    return smooth_common(model, params, n_params, mode, T, N, x_trj, u_trj, a, false, sums, &out,
                         workspace, workspace_bytes, stream);
}

int irs_smooth_rng(int model, const double* params, int n_params, int mode, int T, int N,
                   const double* x_trj, const double* u_trj, const double* std_x, const double* std_u,
                   uint64_t seed, uint32_t iter, double* sums, double* At, double* Bt, double* ct,
                   int* info, void* workspace, size_t workspace_bytes, void* stream) {
    SmoothArgs a;
    memset(&a, 0, sizeof(a));
    int rc = fill_rng(model, mode, std_x, std_u, seed, iter, 0, a);
    if (rc != IRS_OK) return rc;
    SmoothOut out{At, Bt, ct, info, (long long)N};
    return smooth_common(model, params, n_params, mode, T, N, x_trj, u_trj, a, true, sums, &out,
                         workspace, workspace_bytes, stream);
}

int irs_smooth_run(const irs_smooth_call* c, void* stream) {
    IRS_CHECK_ARG(c != nullptr, "null call struct");
    SmoothArgs a;
    memset(&a, 0, sizeof(a));
    int rc;
    if (c->use_rng) {
        rc = fill_rng(c->model, c->mode, c->std_x, c->std_u, c->seed, c->iter, c->sample_offset, a);
    } else {
        rc = check_samples(c->dx, c->du, c->model, c->mode);
        a.dx = c->dx; a.du = c->du;
    }
    if (rc != IRS_OK) return rc;
    SmoothOut out{c->At, c->Bt, c->ct, c->info, c->n_total};
    return smooth_common(c->model, c->params, c->n_params, c->mode, c->T, c->N, c->x_trj, c->u_trj, a,
                         c->use_rng != 0, c->sums, c->At ? &out : nullptr, c->workspace,
                         c->workspace_bytes, stream);
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
    else if (n == 5 && m == 2) hipLaunchKernelGGL((rng_samples_kernel<5, 2>), grid, block, 0, st, dx, du, a);
    else if (n == 6 && m == 2) hipLaunchKernelGGL((rng_samples_kernel<6, 2>), grid, block, 0, st, dx, du, a);
    else if (n == 7 && m == 4) hipLaunchKernelGGL((rng_samples_kernel<7, 4>), grid, block, 0, st, dx, du, a);
    else { irs_set_error("irs_rng_samples: unsupported (n,m)=(%d,%d)", n, m); return IRS_ERR_UNSUPPORTED; }
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_smooth_finalize(int model, const double* params, int n_params, int mode, int T,
                        long long N_total, const double* x_trj, const double* u_trj,
                        const double* sums, double* At, double* Bt, double* ct, int* info,
                        void* stream) {
    return irs_smooth_finalize_ws(model, params, n_params, mode, T, N_total, x_trj, u_trj, sums, At, Bt, ct, info,
                                  nullptr, 0, stream);
}

int irs_smooth_finalize_ws(int model, const double* params, int n_params, int mode, int T,
                           long long N_total, const double* x_trj, const double* u_trj,
                           const double* sums, double* At, double* Bt, double* ct, int* info,
                           const void* workspace, size_t workspace_bytes, void* stream) {
    IRS_CHECK_ARG(T > 0 && N_total > 0, "T and N_total must be positive");
    IRS_CHECK_ARG(workspace == nullptr || workspace_bytes >= (size_t)kCounterBytes + fnom_bytes(T),
                  "workspace too small to hold the nominal steps");
    IRS_CHECK_ARG(x_trj && u_trj && sums && At && Bt && ct && info, "null pointer");
    IRS_CHECK_ARG(mode >= 0 && mode <= 2, "unknown smoothing mode");
    SmoothArgs a;
    memset(&a, 0, sizeof(a));
    int rc = irs_load_params(model, params, n_params, &a.p);
    if (rc != IRS_OK) return rc;
    a.x_trj = x_trj; a.u_trj = u_trj; a.sums = const_cast<double*>(sums);
    a.At = At; a.Bt = Bt; a.ct = ct; a.info = info;
    a.n_total = (double)N_total; a.T = T;
    if (workspace != nullptr)
        a.fnom = reinterpret_cast<double*>(static_cast<char*>(const_cast<void*>(workspace)) + kCounterBytes);
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
