// TV-LQR (irs_lqr/tv_lqr.py:30-145 with inactive bounds) as a backward Riccati pass,
// and the forward loop of IrsLqr.local_descent (irs_lqr/irs_lqr.py:169-184).
//
// Both are T-step sequential chains on matrices of order n <= 32: latency-bound, not
// bandwidth- or FLOP-bound.  Each runs as ONE wave (64 lanes) with its state in LDS
// (f64): no workgroup barriers (a single wave's LDS traffic is ordered), no
// inter-workgroup synchronisation, no host round trips.  Lanes split the output
// elements of every small product; the next step's A,B,c (backward) or K,k (forward)
// are prefetched into registers while the current step computes.
#include "irs_common.hpp"

namespace {

constexpr int kMaxN = 32;
constexpr int kMaxM = 16;

// Orders ONE wave's LDS traffic: the LDS executes a wave's operations in issue order,
// so this only has to stop the compiler from moving them; it deliberately does not
// wait for outstanding global loads (the prefetches stay in flight).
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

struct RiccatiArgs {
    const double* At; const double* Bt; const double* ct;
    const double* Q; const double* Qd; const double* R;
    const double* xd;
    double* K; double* k;
    int* info;
    double alpha;
    int n, m, T;
};

// ------------------------------------------------------------------ compile-time sizes
template <int N, int M>
struct RiccatiLds {
    double P[N * N], A[N * N], Acl[N * N], W[N * N], Q[N * N];
    double B[N * M], PB[N * M], Kt[M * N], G1[M * N];
    double H[M * M], R[M * M];
    double p[N], c[N], qv[N], xd[N], g[M], kt[M];
    int bad;
    unsigned char ti[N * (N + 1) / 2], tj[N * (N + 1) / 2];
};

// 1/d to ~1 ulp: hardware reciprocal + two Newton steps (a correctly rounded f64 divide
// is ~40 dependent instructions on the critical path of every Riccati step).
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// Backward pass.  On return K (T,M,N) / k (T,M) are in global memory.
template <int N, int M>
__device__ __forceinline__ void riccati_backward(const RiccatiArgs& a, int lane, RiccatiLds<N, M>& S) {
    constexpr int NN = N * N, NM = N * M, NT = N * (N + 1) / 2;
    constexpr int RA = (NN + 63) / 64, RB = (NM + 63) / 64, RT = (NT + 63) / 64;
    const int T = a.T;

    // only the symmetric parts of Q, Qd, R enter a quadratic form
    for (int q = lane; q < NN; q += 64) {
        int i = q / N, j = q % N;
        S.P[q] = 0.5 * (a.Qd[q] + a.Qd[j * N + i]);
        S.Q[q] = 0.5 * (a.Q[q] + a.Q[j * N + i]);
    }
    for (int q = lane; q < M * M; q += 64) {
        int i = q / M, j = q % M;
        S.R[q] = 0.5 * a.alpha * (a.R[q] + a.R[j * M + i]);
    }
    // (i,j), i <= j, of the q-th upper-triangular element
    for (int q = lane; q < NN; q += 64) {
        int i = q / N, j = q % N;
        if (i <= j) {
            int idx = i * N - i * (i - 1) / 2 + (j - i);
            S.ti[idx] = (unsigned char)i;
            S.tj[idx] = (unsigned char)j;
        }
    }
    if (lane == 0) S.bad = 0;
    // step T-1 operands
    {
        const double* At = a.At + (size_t)(T - 1) * NN;
        const double* Bt = a.Bt + (size_t)(T - 1) * NM;
        for (int q = lane; q < NN; q += 64) S.A[q] = At[q];
        for (int q = lane; q < NM; q += 64) S.B[q] = Bt[q];
        if (lane < N) {
            S.c[lane] = a.ct[(size_t)(T - 1) * N + lane];
            S.xd[lane] = a.xd[(size_t)(T - 1) * N + lane];
        }
    }
    wave_sync();
    if (lane < N) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) s -= S.P[lane * N + j] * a.xd[(size_t)T * N + j];
        S.p[lane] = s;
    }
    wave_sync();

    for (int t = T - 1; t >= 0; --t) {
        // prefetch the operands of step t-1 (independent of this step's arithmetic)
        double ra[RA], rb[RB], rc = 0.0, rxd = 0.0;
        const int tp = t > 0 ? t - 1 : 0;
#pragma unroll
        for (int r = 0; r < RA; ++r) { int q = lane + 64 * r; ra[r] = q < NN ? a.At[(size_t)tp * NN + q] : 0.0; }
#pragma unroll
        for (int r = 0; r < RB; ++r) { int q = lane + 64 * r; rb[r] = q < NM ? a.Bt[(size_t)tp * NM + q] : 0.0; }
        if (lane < N) { rc = a.ct[(size_t)tp * N + lane]; rxd = a.xd[(size_t)tp * N + lane]; }

        // 1. PB = P B ; qv = P c + p
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            int q = lane + 64 * r;
            if (q < NM) {
                int i = q / M, j = q % M;
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < N; ++l) s += S.P[i * N + l] * S.B[l * M + j];
                S.PB[q] = s;
            }
        }
        if (lane < N) {
            double s = S.p[lane];
#pragma unroll
            for (int l = 0; l < N; ++l) s += S.P[lane * N + l] * S.c[l];
            S.qv[lane] = s;
        }
        wave_sync();
        // 2. H = alpha R + B'PB ; G1 = (PB)'A ; g = B'q
        for (int q = lane; q < M * M; q += 64) {
            int i = q / M, j = q % M;
            double s = S.R[q];
#pragma unroll
            for (int l = 0; l < N; ++l) s += S.B[l * M + i] * S.PB[l * M + j];
            S.H[q] = s;
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            int q = lane + 64 * r;
            if (q < NM) {
                int i = q / N, j = q % N;
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < N; ++l) s += S.PB[l * M + i] * S.A[l * N + j];
                S.G1[q] = s;
            }
        }
        if (lane < M) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) s += S.B[l * M + lane] * S.qv[l];
            S.g[lane] = s;
        }
        wave_sync();
        // 3. K = -H^-1 G1, k = -H^-1 g: every lane factors the M x M Hessian in registers
        //    (LDL'), lanes 0..N-1 own a column of G1, lane N owns g.
        {
            double Lm[M][M], Dg[M], Dinv[M];
            bool ok = true;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double dj = S.H[j * M + j];
#pragma unroll
                for (int l = 0; l < j; ++l) dj -= Lm[j][l] * Lm[j][l] * Dg[l];
                ok = ok && (dj > 0.0);
                Dg[j] = dj;
                Dinv[j] = fast_rcp(dj);
#pragma unroll
                for (int i = j + 1; i < M; ++i) {
                    double s = S.H[i * M + j];
#pragma unroll
                    for (int l = 0; l < j; ++l) s -= Lm[i][l] * Lm[j][l] * Dg[l];
                    Lm[i][j] = s * Dinv[j];
                }
            }
            if (!ok && lane == 0 && S.bad == 0) S.bad = t + 1;
            if (lane <= N) {
                double y[M];
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    double s = (lane < N) ? S.G1[i * N + (lane < N ? lane : 0)] : S.g[i];
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
                for (int i = 0; i < M; ++i) {
                    if (lane < N) {
                        S.Kt[i * N + lane] = -y[i];
                        a.K[((size_t)t * M + i) * N + lane] = -y[i];
                    } else {
                        S.kt[i] = -y[i];
                        a.k[(size_t)t * M + i] = -y[i];
                    }
                }
            }
        }
        wave_sync();
        // 4. Acl = A + B K
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            int q = lane + 64 * r;
            if (q < NN) {
                int i = q / N, j = q % N;
                double s = S.A[q];
#pragma unroll
                for (int l = 0; l < M; ++l) s += S.B[i * M + l] * S.Kt[l * N + j];
                S.Acl[q] = s;
            }
        }
        wave_sync();
        // 5. W = P Acl ; p_new = Acl' q - Q xd_t
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            int q = lane + 64 * r;
            if (q < NN) {
                int i = q / N, j = q % N;
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < N; ++l) s += S.P[i * N + l] * S.Acl[l * N + j];
                S.W[q] = s;
            }
        }
        double pnew = 0.0;
        if (lane < N) {
#pragma unroll
            for (int l = 0; l < N; ++l) pnew += S.Acl[l * N + lane] * S.qv[l] - S.Q[lane * N + l] * S.xd[l];
        }
        wave_sync();
        // 6. P = Q + A'W.  A'P(A+BK) is symmetric: only the upper triangle is computed
        //    and mirrored (exact symmetry, half the work).  Then rotate in the prefetched
        //    operands of step t-1.
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            int q = lane + 64 * r;
            if (q < NT) {
                int i = S.ti[q], j = S.tj[q];
                double s = S.Q[i * N + j];
#pragma unroll
                for (int l = 0; l < N; ++l) s += S.A[l * N + i] * S.W[l * N + j];
                S.P[i * N + j] = s;
                S.P[j * N + i] = s;
            }
        }
        if (lane < N) S.p[lane] = pnew;
        wave_sync();
#pragma unroll
        for (int r = 0; r < RA; ++r) { int q = lane + 64 * r; if (q < NN) S.A[q] = ra[r]; }
#pragma unroll
        for (int r = 0; r < RB; ++r) { int q = lane + 64 * r; if (q < NM) S.B[q] = rb[r]; }
        if (lane < N) { S.c[lane] = rc; S.xd[lane] = rxd; }
        wave_sync();
    }
    if (lane == 0) a.info[0] = S.bad;
}

// Tiny systems (N <= 4): the whole recursion lives in registers, every lane carrying
// the same values (uniform addresses -> scalar loads); no LDS, no waits but the loads'.
template <int N, int M>
__device__ __forceinline__ void riccati_backward_reg(const RiccatiArgs& a, int lane) {
    const int T = a.T;
    double P[N][N], p[N], Q[N][N], R[M][M];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) { P[i][j] = a.Qd[i * N + j]; Q[i][j] = a.Q[i * N + j]; }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) R[i][j] = a.alpha * a.R[i * M + j];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) s -= P[i][j] * a.xd[(size_t)T * N + j];
        p[i] = s;
    }
    double An[N][N], Bn[N][M], cn[N], xdn[N];
    auto fetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < N; ++j) An[i][j] = a.At[((size_t)t * N + i) * N + j];
#pragma unroll
            for (int j = 0; j < M; ++j) Bn[i][j] = a.Bt[((size_t)t * N + i) * M + j];
            cn[i] = a.ct[(size_t)t * N + i];
            xdn[i] = a.xd[(size_t)t * N + i];
        }
    };
    fetch(T - 1);
    int bad = 0;
    for (int t = T - 1; t >= 0; --t) {
        double A[N][N], B[N][M], c[N], xd[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < N; ++j) A[i][j] = An[i][j];
#pragma unroll
            for (int j = 0; j < M; ++j) B[i][j] = Bn[i][j];
            c[i] = cn[i];
            xd[i] = xdn[i];
        }
        fetch(t > 0 ? t - 1 : 0);
        double PB[N][M], qv[N], H[M][M], G1[M][N], g[M];
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < N; ++l) s += P[i][l] * B[l][j];
                PB[i][j] = s;
            }
            double s = p[i];
#pragma unroll
            for (int l = 0; l < N; ++l) s += P[i][l] * c[l];
            qv[i] = s;
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double s = R[i][j];
#pragma unroll
                for (int l = 0; l < N; ++l) s += B[l][i] * PB[l][j];
                H[i][j] = s;
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < N; ++l) s += PB[l][i] * A[l][j];
                G1[i][j] = s;
            }
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) s += B[l][i] * qv[l];
            g[i] = s;
        }
        // LDL' of H, then K = -H^-1 G1, k = -H^-1 g
        double Lm[M][M], Dg[M], Dinv[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double dj = H[j][j];
#pragma unroll
            for (int l = 0; l < j; ++l) dj -= Lm[j][l] * Lm[j][l] * Dg[l];
            if (!(dj > 0.0) && bad == 0) bad = t + 1;
            Dg[j] = dj;
            Dinv[j] = fast_rcp(dj);
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
                double s = H[i][j];
#pragma unroll
                for (int l = 0; l < j; ++l) s -= Lm[i][l] * Lm[j][l] * Dg[l];
                Lm[i][j] = s * Dinv[j];
            }
        }
        double K[M][N], kk[M];
#pragma unroll
        for (int col = 0; col <= N; ++col) {
            double y[M];
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double s = col < N ? G1[i][col < N ? col : 0] : g[i];
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
            for (int i = 0; i < M; ++i) {
                if (col < N) K[i][col < N ? col : 0] = -y[i];
                else kk[i] = -y[i];
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < M; ++i) {
#pragma unroll
                for (int j = 0; j < N; ++j) a.K[((size_t)t * M + i) * N + j] = K[i][j];
                a.k[(size_t)t * M + i] = kk[i];
            }
        }
        double Acl[N][N], W[N][N], pn[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double s = A[i][j];
#pragma unroll
                for (int l = 0; l < M; ++l) s += B[i][l] * K[l][j];
                Acl[i][j] = s;
            }
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < N; ++l) s += P[i][l] * Acl[l][j];
                W[i][j] = s;
            }
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) s += Acl[l][i] * qv[l] - Q[i][l] * xd[l];
            pn[i] = s;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double s = 0.0, s2 = 0.0;
#pragma unroll
                for (int l = 0; l < N; ++l) { s += A[l][i] * W[l][j]; s2 += A[l][j] * W[l][i]; }
                P[i][j] = Q[i][j] + 0.5 * (s + s2);
            }
            p[i] = pn[i];
        }
    }
    if (lane == 0) a.info[0] = bad;
}

// ------------------------------------------------------------------ matrix-core path
// 4 < N <= 15, M <= 4 (quadrotor 12/4, bicycle 5/2, three_cart 6/2): the recursion in
// homogeneous coordinates z = [x; 1],
//     A~ = [A c; 0 1],  B~ = [B; 0],  P~ = [P p; p' r],  Q~_t = [Q -Q xd_t; -(Q xd_t)' *],
//     K~ = -(aR + B~'P~B~)^-1 B~'P~A~ = [K | k],     P~ <- Q~_t + A~'P~(A~ + B~K~),
// carries the affine terms (c, p, k) inside ONE 16x16 tile, and every product chains
// through v_mfma_f64_16x16x4_f64 in registers: the C/D layout of that instruction
// (col = lane&15, row = (lane>>4) + 4*reg) IS its B-operand layout (k = (lane>>4) + 4*step),
// and the A-operand layout of the TRANSPOSE -- so P~ (symmetric), A~' and B~' need no
// data movement at all.  No LDS traffic, no waits inside a step except the 4x4 solve.
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kQxMax = 4096;          // doubles of LDS for Q xd_t, t = 0..T

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// -DIRS_RIC_STAMPS (tuning builds only): s_memtime totals per phase of the matrix-core recursion
#ifdef IRS_RIC_STAMPS
__device__ long long ric_stamps[8];
#define RIC_MARK(k) do { const long long n_ = __builtin_amdgcn_s_memtime(); rs_[k] += n_ - rm_; rm_ = n_; } while (0)
#else
#define RIC_MARK(k) do {} while (0)
#endif

template <int N, int M>
__device__ __forceinline__ void riccati_backward_mfma(const RiccatiArgs& a, int lane, double* qx) {
    static_assert(N + 1 <= 16 && M <= 4, "one 16x16 tile, gains in accumulator register 0");
    constexpr int KA = (N + 1 + 3) / 4;      // k-steps over the augmented dimension
    constexpr int KB = (N + 3) / 4;          // k-steps when the operand's rows >= N are zero
    const int T = a.T;
    const int col = lane & 15, rg = lane >> 4;
    auto qs = [&](const double* Qm, int i, int j) { return 0.5 * (Qm[i * N + j] + Qm[j * N + i]); };

    // qx[t][i] = (Q xd_t)_i for every t, once, in parallel
    for (int idx = lane; idx < (T + 1) * N; idx += 64) {
        const int t = idx / N, i = idx % N;
        double s = 0.0;
        for (int j = 0; j < N; ++j) s += qs(a.Q, i, j) * a.xd[(size_t)t * N + j];
        qx[idx] = s;
    }
    double Rr[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) Rr[i][j] = 0.5 * a.alpha * (a.R[i * M + j] + a.R[j * M + i]);
    v4d Qc, P;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = rg + 4 * r;
        Qc[r] = (row < N && col < N) ? qs(a.Q, row, col) : 0.0;
        // P~_T = [Qd, -Qd xd_T; ., 0]
        double v = 0.0;
        if (row < N && col < N) v = qs(a.Qd, row, col);
        else if ((col == N && row < N) || (row == N && col < N)) {
            const int i = col == N ? row : col;
            for (int j = 0; j < N; ++j) v -= qs(a.Qd, i, j) * a.xd[(size_t)T * N + j];
        }
        P[r] = v;
    }
    // Which array (and where) each of this lane's nine operand elements comes from is fixed: resolved ONCE into
    // (pointer, stride per time step, constant, is-a-load), so that a step's loads are nine unconditional global loads
    // (an element that is a constant reads a.ct[0] and discards it) and its Q~ column two plain LDS reads -- the
    // per-element branches this replaces (exec-mask regions, each with its own wait) were a third of the step.
    typedef const __attribute__((address_space(1))) double* gptr;
    gptr lp[9];
    int lstr[9];
    double lc[9];
    bool lld[9];
    {
        const gptr gA = (gptr)a.At, gB = (gptr)a.Bt, gc_ = (gptr)a.ct;
#pragma unroll
        for (int e = 0; e < 9; ++e) { lp[e] = gc_; lstr[e] = 0; lc[e] = 0.0; lld[e] = false; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rg + 4 * r;
            if (row < N) {
                if (col < N) { lp[r] = gA + row * N + col; lstr[r] = N * N; lld[r] = true; }
                else if (col == N) { lp[r] = gc_ + row; lstr[r] = N; lld[r] = true; }
                if (col < M) { lp[4 + r] = gB + row * M + col; lstr[4 + r] = N * M; lld[4 + r] = true; }
            } else if (row == N && col == N) {
                lc[r] = 1.0;
            }
        }
        if (col < N && rg < M) { lp[8] = gB + col * M + rg; lstr[8] = N * M; lld[8] = true; }
    }
    // raw loads now, selection (load or constant) at the point of use one step later: a select right behind its load
    // would park the wave on the load's latency at the top of every step
    auto load_raw = [&](int t, double* v) {
#pragma unroll
        for (int e = 0; e < 9; ++e) v[e] = lp[e][(long)t * lstr[e]];
    };
    auto finish_raw = [&](const double* v, v4d& Ab, v4d& Bb, double& Ba) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Ab[r] = lld[r] ? v[r] : lc[r];
            Bb[r] = lld[4 + r] ? v[4 + r] : 0.0;
        }
        Ba = lld[8] ? v[8] : 0.0;
    };
    // the -Q xd_t entries of the stage-cost tile: element (row < N, col N) and its mirror
    int qidx[4];
    bool qhas[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = rg + 4 * r;
        qhas[r] = (col == N && row < N) || (row == N && col < N);
        qidx[r] = qhas[r] ? (col == N ? row : col) : 0;
    }
    // where this lane's gain element goes: K[t][rg][col], k[t][rg] (col == N), or nowhere
    double* kst = (rg < M && col < N) ? a.K + rg * N + col : ((rg < M && col == N) ? a.k + rg : nullptr);
    const int kstr = (rg < M && col < N) ? M * N : M;
    double rawn[9], qn[4];
    load_raw(T - 1, rawn);
    wave_sync();                                     // qx visible
    // (the Q xd entries travel with the prefetch too: read one step ahead and carried across the back edge -- read
    // at the point of use, the compiler turns the selection below into four exec-masked LDS reads, each with its wait)
#pragma unroll
    for (int r = 0; r < 4; ++r) qn[r] = qx[(T - 1) * N + qidx[r]];
    bool ok = true;
    int bad_t = 0;
#ifdef IRS_RIC_STAMPS
    long long rs_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rm_ = __builtin_amdgcn_s_memtime();
#endif
    for (int t = T - 1; t >= 0; --t) {
#ifdef IRS_RIC_STAMPS
        {   // how long the step waits for its prefetched operands
            const long long w0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rs_[5] += __builtin_amdgcn_s_memtime() - w0;
            rm_ = __builtin_amdgcn_s_memtime();
        }
#endif
        v4d Ab, Bb;
        double Ba;
        finish_raw(rawn, Ab, Bb, Ba);
        load_raw(t > 0 ? t - 1 : 0, rawn);           // prefetch
        v4d Qt;
#pragma unroll
        for (int r = 0; r < 4; ++r) Qt[r] = qhas[r] ? -qn[r] : Qc[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) qn[r] = qx[(t > 0 ? t - 1 : 0) * N + qidx[r]];
        // D1 = P~ A~ , D2 = P~ B~
        v4d D1 = {0, 0, 0, 0}, D2 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KA; ++s) {
            D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[s], Ab[s], D1, 0, 0, 0);
            D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[s], Bb[s], D2, 0, 0, 0);
        }
        // G~ = B~' D1 (rows < M: [B'PA | B'(Pc+p)]),  Hh = B~' D2 = B'PB
        v4d G = {0, 0, 0, 0}, Hh = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            G = __builtin_amdgcn_mfma_f64_16x16x4f64(Bb[s], D1[s], G, 0, 0, 0);
            Hh = __builtin_amdgcn_mfma_f64_16x16x4f64(Bb[s], D2[s], Hh, 0, 0, 0);
        }
        asm volatile("" :: "v"(G[0]), "v"(Hh[0]));
        RIC_MARK(0);                                  // head: Q column, prefetch issue, 14 products
        // M x M Hessian to every lane (element (i,j) sits in register 0 of lane 16 i + j)
        double H[M][M];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) H[i][j] = Rr[i][j] + readlane_f64(Hh[0], 16 * i + j);
        double Lm[M][M], Dg[M], Dinv[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            double dj = H[j][j];
#pragma unroll
            for (int l = 0; l < j; ++l) dj -= Lm[j][l] * Lm[j][l] * Dg[l];
            if (!(dj > 0.0) && ok) { ok = false; bad_t = t + 1; }
            Dg[j] = dj;
            Dinv[j] = fast_rcp(dj);
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
                double s = H[i][j];
#pragma unroll
                for (int l = 0; l < j; ++l) s -= Lm[i][l] * Lm[j][l] * Dg[l];
                Lm[i][j] = s * Dinv[j];
            }
        }
        asm volatile("" :: "v"(Dinv[0]), "v"(Dinv[M - 1]));
        RIC_MARK(1);                                  // gather of H, LDL'
        // this lane's column of G~ (its M entries live in lanes 16 i + col), solve, keep row rg
        double y[M];
#pragma unroll
        for (int i = 0; i < M; ++i) y[i] = __shfl(G[0], 16 * i + col, 64);
#pragma unroll
        for (int i = 0; i < M; ++i) {
#pragma unroll
            for (int l = 0; l < i; ++l) y[i] -= Lm[i][l] * y[l];
        }
#pragma unroll
        for (int i = M - 1; i >= 0; --i) {
            double s = y[i] * Dinv[i];
#pragma unroll
            for (int l = i + 1; l < M; ++l) s -= Lm[l][i] * y[l];
            y[i] = s;
        }
        double Kb = 0.0;                              // K~[rg][col] = [K | k]
#pragma unroll
        for (int i = 0; i < M; ++i) Kb = (i == rg) ? -y[i] : Kb;
        if (rg >= M || col > N) Kb = 0.0;
        if (kst != nullptr) kst[(size_t)t * kstr] = Kb;
        asm volatile("" :: "v"(Kb));
        RIC_MARK(2);                                  // column gather, substitutions, gain, store
        // A~cl = A~ + B~ K~ ; W = P~ A~cl ; Joseph form  P~ <- Q~_t + K~'(aR)K~ + A~cl' W.
        // (The plain form Q~ + A~'W uses the open-loop A~ on one side: the antisymmetric
        // rounding error of P~ -- which the transposed-operand trick turns into a sign flip --
        // then grows ~3x per step on the quadrotor.  With A~cl on both sides it contracts.)
        v4d Acl = __builtin_amdgcn_mfma_f64_16x16x4f64(Ba, Kb, Ab, 0, 0, 0);
        v4d W = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KA; ++s) W = __builtin_amdgcn_mfma_f64_16x16x4f64(P[s], Acl[s], W, 0, 0, 0);
        double RKb = 0.0;                             // (aR K~)[rg][col]
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double rk = 0.0;
#pragma unroll
            for (int q = 0; q < M; ++q) rk -= Rr[i][q] * y[q];
            RKb = (i == rg) ? rk : RKb;
        }
        if (rg >= M || col > N) RKb = 0.0;
        v4d Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(Kb, RKb, Qt, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < KA; ++s) Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(Acl[s], W[s], Pn, 0, 0, 0);
        P = Pn;
        asm volatile("" :: "v"(P[0]));
        RIC_MARK(3);                                  // closed loop, W, Joseph form: 10 products
    }
#ifdef IRS_RIC_STAMPS
    if (lane == 0) {
        for (int k_ = 0; k_ < 6; ++k_) ric_stamps[k_] = rs_[k_];
        ric_stamps[4] = T;
    }
#endif
    if (lane == 0) a.info[0] = bad_t;
}

template <int N, int M>
__device__ __forceinline__ void riccati_backward_any(const RiccatiArgs& a, int lane, RiccatiLds<N, M>& S,
                                                     double* qx) {
    if constexpr (N <= 4 && M <= 2) {
        riccati_backward_reg<N, M>(a, lane);
    } else if constexpr (N + 1 <= 16 && M <= 4) {
        if ((a.T + 1) * N <= kQxMax) riccati_backward_mfma<N, M>(a, lane, qx);
        else riccati_backward<N, M>(a, lane, S);
    } else {
        riccati_backward<N, M>(a, lane, S);
    }
}

template <int N, int M>
__global__ __launch_bounds__(64) void riccati_kernel_t(RiccatiArgs a) {
    __shared__ RiccatiLds<N, M> S;
    __shared__ double qx[(N > 4 && N + 1 <= 16 && M <= 4) ? kQxMax : 1];
    riccati_backward_any<N, M>(a, threadIdx.x, S, qx);
}

// ------------------------------------------------------------------ generic runtime sizes
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
    wave_sync();
    if (lane < n) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s -= P[lane * n + j] * a.xd[(size_t)T * n + j];
        pv[lane] = s;
    }
    wave_sync();

    for (int t = T - 1; t >= 0; --t) {
        const double* At = a.At + (size_t)t * n * n;
        const double* Bt = a.Bt + (size_t)t * n * m;
        const double* ct = a.ct + (size_t)t * n;
        for (int q = lane; q < n * n; q += 64) A[q] = At[q];
        for (int q = lane; q < n * m; q += 64) B[q] = Bt[q];
        wave_sync();
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
        wave_sync();
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
        wave_sync();
        for (int j = 0; j < m; ++j) {
            double djj = Hm[j * ldh + j];
            if (!(djj > 0.0)) {
                if (lane == 0 && bad == 0) bad = t + 1;
                djj = 1.0;
            }
            double l = sqrt(djj);
            wave_sync();
            if (lane == j) Hm[j * ldh + j] = l;
            if (lane > j && lane < m) Hm[lane * ldh + j] /= l;
            wave_sync();
            for (int q = lane; q < m * m; q += 64) {
                int r = q / m, c = q % m;
                if (c > j && r >= c) Hm[r * ldh + c] -= Hm[r * ldh + j] * Hm[c * ldh + j];
            }
            wave_sync();
        }
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
            wave_sync();
            for (int i = 0; i < m; ++i) {
                if (lane < n) Kt[i * n + lane] = -y[i];
                else kt[i] = -y[i];
            }
        }
        wave_sync();
        for (int q = lane; q < m * n; q += 64) a.K[(size_t)t * m * n + q] = Kt[q];
        if (lane < m) a.k[(size_t)t * m + lane] = kt[lane];
        for (int q = lane; q < n * n; q += 64) {
            int i = q / n, j = q % n;
            double s = A[q];
            for (int l = 0; l < m; ++l) s += B[i * m + l] * Kt[l * n + j];
            Acl[q] = s;
        }
        wave_sync();
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
        wave_sync();
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
        wave_sync();
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
    wave_sync();
    for (int t = 0; t < T; ++t) {
        if (lane < m) {
            double s = k[(size_t)t * m + lane];
            for (int j = 0; j < n; ++j) s += K[((size_t)t * m + lane) * n + j] * x[j];
            u[lane] = s;
            us[(size_t)t * m + lane] = s;
        }
        wave_sync();
        double xn = 0.0;
        if (lane < n) {
            xn = ct[(size_t)t * n + lane];
            for (int j = 0; j < n; ++j) xn += At[((size_t)t * n + lane) * n + j] * x[j];
            for (int j = 0; j < m; ++j) xn += Bt[((size_t)t * n + lane) * m + j] * u[j];
        }
        wave_sync();
        if (lane < n) { x[lane] = xn; xs[(size_t)(t + 1) * n + lane] = xn; }
        wave_sync();
    }
}

// ------------------------------------------------------------------ rollouts on the true dynamics
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

template <class Model>
struct RolloutLds {
    double Q[Model::NX * Model::NX];
    double R[Model::NU * Model::NU];
    double K[Model::NU * Model::NX];
    double k[Model::NU];
    double xd[Model::NX];
    double u[Model::NU];
};

// Closed-loop (K != null) or open-loop (K == null, u = u_in) rollout on the TRUE
// dynamics + evaluate_cost.  Sequential in t: every lane of the single wave carries
// the same state in registers (no divergence, no broadcasts); lane 0 stores.  The
// next step's gains / references are prefetched into registers, staged through LDS.
template <class Model>
__device__ __forceinline__ void rollout_device(const ModelParams& p, int T, const double* K, const double* k,
                                               const double* u_in, const double* x0, const double* Q,
                                               const double* R, const double* xd_trj, double* x_out,
                                               double* u_out, double* cost_out, int lane,
                                               RolloutLds<Model>& S) {
    constexpr int n = Model::NX, m = Model::NU, MN = m * n;
    constexpr int RK = (MN + 63) / 64;
    static_assert(m + n <= 64, "state too large for this staging scheme");
    if constexpr (n <= 4 && m <= 2) {
        // tiny model: everything in registers, uniform (scalar) loads one step ahead
        double Qr[n][n], Rr[m][m], x[n], u[m], xn[n];
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < n; ++j) Qr[i][j] = Q[i * n + j];
#pragma unroll
        for (int i = 0; i < m; ++i)
#pragma unroll
            for (int j = 0; j < m; ++j) Rr[i][j] = R[i * m + j];
#pragma unroll
        for (int i = 0; i < n; ++i) x[i] = x0[i];
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < n; ++i) x_out[i] = x[i];
        }
        double Kn[m][n], kn[m], xdn[n];
        auto fetch = [&](int t, bool gains) {
            if (gains) {
#pragma unroll
                for (int i = 0; i < m; ++i) {
                    if (K != nullptr) {
#pragma unroll
                        for (int j = 0; j < n; ++j) Kn[i][j] = K[((size_t)t * m + i) * n + j];
                        kn[i] = k[(size_t)t * m + i];
                    } else {
                        kn[i] = u_in[(size_t)t * m + i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) xdn[i] = xd_trj[(size_t)t * n + i];
        };
        fetch(0, true);
        double cost = 0.0;
        unsigned warm_set = ~0u;      // active set of the previous contact step (exact step QPs only)
        for (int t = 0; t < T; ++t) {
            double e[n];
#pragma unroll
            for (int i = 0; i < m; ++i) {
                double s = kn[i];
                if (K != nullptr) {
#pragma unroll
                    for (int j = 0; j < n; ++j) s += Kn[i][j] * x[j];
                }
                u[i] = s;
            }
#pragma unroll
            for (int i = 0; i < n; ++i) e[i] = x[i] - xdn[i];
            fetch(t + 1, t + 1 < T);      // xd_trj has a row T; the gains do not
#pragma unroll
            for (int i = 0; i < n; ++i) {
                double r = 0.0;
#pragma unroll
                for (int j = 0; j < n; ++j) r += Qr[i][j] * e[j];
                cost += e[i] * r;
            }
#pragma unroll
            for (int i = 0; i < m; ++i) {
                double r = 0.0;
#pragma unroll
                for (int j = 0; j < m; ++j) r += Rr[i][j] * u[j];
                cost += u[i] * r;
            }
            irs_step_along<Model>(p, x, u, xn, &warm_set);
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
        // terminal term uses Q, not Qd (irs_lqr/irs_lqr.py:135-136); xdn holds xd_T here
        double eT[n];
#pragma unroll
        for (int i = 0; i < n; ++i) eT[i] = x[i] - xdn[i];
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) r += Qr[i][j] * eT[j];
            cost += eT[i] * r;
        }
        if (lane == 0) cost_out[0] = cost;
        return;
    }
    for (int q = lane; q < n * n; q += 64) S.Q[q] = Q[q];
    for (int q = lane; q < m * m; q += 64) S.R[q] = R[q];
    double x[n], u[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = x0[i];
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < n; ++i) x_out[i] = x[i];
    }
    // operands of step 0
    double rk[RK], rv = 0.0;
    auto fetch = [&](int t) {
        if (K != nullptr) {
#pragma unroll
            for (int r = 0; r < RK; ++r) { int q = lane + 64 * r; rk[r] = q < MN ? K[(size_t)t * MN + q] : 0.0; }
            if (lane < m) rv = k[(size_t)t * m + lane];
        } else {
            if (lane < m) rv = u_in[(size_t)t * m + lane];
        }
        if (lane >= m && lane < m + n) rv = xd_trj[(size_t)t * n + (lane - m)];
    };
    fetch(0);
    // Lane-distributed bookkeeping: lane i < m owns u_i and the i-th row of u'Ru, lane
    // i < n owns the i-th row of e'Qe; each lane accumulates its share of the cost over
    // all t and the wave is reduced ONCE at the end.  Only the dynamics step is redundant.
    double cost = 0.0;
    unsigned warm_set = ~0u;
    auto pick = [&](const double* v, int len) {       // v[lane] without dynamic register indexing
        double r = v[0];
#pragma unroll
        for (int i = 1; i < (n > m ? n : m); ++i) r = (i < len && i == lane) ? v[i] : r;
        return r;
    };
    auto state_cost = [&](const double* xdv) {        // lane i < n: e_i * (Q e)_i
        if (lane < n) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) r += S.Q[lane * n + j] * (x[j] - xdv[j]);
            cost += (pick(x, n) - xdv[lane]) * r;
        }
    };
    for (int t = 0; t < T; ++t) {
        wave_sync();
        if (K != nullptr) {
#pragma unroll
            for (int r = 0; r < RK; ++r) { int q = lane + 64 * r; if (q < MN) S.K[q] = rk[r]; }
        }
        if (lane < m) S.k[lane] = rv;
        if (lane >= m && lane < m + n) S.xd[lane - m] = rv;
        wave_sync();
        fetch(t + 1 < T ? t + 1 : t);
        if (lane < m) {
            double s = S.k[lane];
            if (K != nullptr) {
#pragma unroll
                for (int j = 0; j < n; ++j) s += S.K[lane * n + j] * x[j];
            }
            S.u[lane] = s;
        }
        state_cost(S.xd);
        wave_sync();
#pragma unroll
        for (int i = 0; i < m; ++i) u[i] = S.u[i];
        if (lane < m) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) r += S.R[lane * m + j] * u[j];
            cost += pick(u, m) * r;
        }
        irs_step_along<Model>(p, x, u, xn, &warm_set);
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
    wave_sync();
    if (lane < n) S.xd[lane] = xd_trj[(size_t)T * n + lane];
    wave_sync();
    state_cost(S.xd);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) cost += __shfl_xor(cost, sft, 64);
    if (lane == 0) cost_out[0] = cost;
}

template <class Model>
__global__ __launch_bounds__(64) void rollout_kernel(ModelParams p, int T, const double* K,
                                                     const double* k, const double* u_in,
                                                     const double* x0, const double* Q,
                                                     const double* R, const double* xd_trj,
                                                     double* x_out, double* u_out, double* cost_out) {
    __shared__ RolloutLds<Model> S;
    rollout_device<Model>(p, T, K, k, u_in, x0, Q, R, xd_trj, x_out, u_out, cost_out, threadIdx.x, S);
}

// One launch for the whole of IrsLqr.local_descent after get_TV_matrices
// (irs_lqr/irs_lqr.py:169-184) + evaluate_cost: backward Riccati, then the closed-loop
// rollout of the policy it just wrote.
template <class Model>
__global__ __launch_bounds__(64) void descent_kernel(ModelParams p, RiccatiArgs a, const double* x0,
                                                     double* x_new, double* u_new, double* cost_out,
                                                     const int* smooth_info, int* row) {
    __shared__ RiccatiLds<Model::NX, Model::NU> S;
    __shared__ RolloutLds<Model> S2;
    const int lane = threadIdx.x;
    __shared__ double qx[(Model::NX > 4 && Model::NX + 1 <= 16 && Model::NU <= 4) ? kQxMax : 1];
    riccati_backward_any<Model::NX, Model::NU>(a, lane, S, qx);
    // the gains were stored by this very wave: drain the stores, then re-read them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rollout_device<Model>(p, a.T, a.K, a.k, nullptr, x0, a.Q, a.R, a.xd, x_new, u_new, cost_out, lane, S2);
    // fused iterate (iterate.hip), no bounds: this descent's row of the info history -- [0] Riccati info,
    // [1] timesteps whose smoothing solve failed, the rest zero -- without a launch of its own
    if (row != nullptr) {
        int bad = 0;
        if (smooth_info != nullptr)
            for (int t = lane; t < a.T; t += 64) bad += smooth_info[t] != 0 ? 1 : 0;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) bad += __shfl_xor(bad, sft, 64);
        if (lane == 0) {
            row[0] = a.info[0];
            row[1] = bad;
#pragma unroll
            for (int i = 2; i < 8; ++i) row[i] = 0;
        }
    }
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

// IrsLqrZeroOrder.compute_least_squares (irs_lqr/irs_lqr_zero_order.py:27-36) stand-alone:
//   ABhat = lstsq([dx du] (N x d), deltaf (N x n))[0]'   ->   A (n x n), B (n x m)
// One workgroup: the Gram matrix Z'Z and the cross term Z'dF are accumulated in f64 (every thread owns
// entries and walks the N rows), then one wave solves the Jacobi-scaled normal equations by Cholesky --
// the same solve the sample pass ends with (smooth.hip, finalize_timestep), for supplied samples and
// without a model.  info = 0, or the 1-based pivot that failed (rank-deficient design), or d + 1 for a
// non-finite entry.
__global__ __launch_bounds__(256) void lstsq_kernel(int n, int m, int N, const double* Z, const double* dF,
                                                    double* A, double* B, int* info) {
    constexpr int D = kMaxN + kMaxM;
    __shared__ double G[D][D + 1];
    __shared__ double H[D][kMaxN];
    __shared__ double sc[D];
    __shared__ int bad;
    const int d = n + m, tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int q = tid; q < d * d; q += 256) {
        const int i = q / d, j = q % d;
        if (j < i) continue;
        double s = 0.0;
        for (int r = 0; r < N; ++r) s = fma(Z[(size_t)r * d + i], Z[(size_t)r * d + j], s);
        G[i][j] = s;
        G[j][i] = s;
    }
    for (int q = tid; q < d * n; q += 256) {
        const int i = q / n, k = q % n;
        double s = 0.0;
        for (int r = 0; r < N; ++r) s = fma(Z[(size_t)r * d + i], dF[(size_t)r * n + k], s);
        H[i][k] = s;
    }
    __syncthreads();
    if (tid >= 64) return;
    const int lane = tid;
    auto wsync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    if (lane < d) {
        const double g = G[lane][lane];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(g);
        if ((bits & 0x7ff0000000000000ull) == 0x7ff0000000000000ull) bad = d + 1;
        else if (!(g > 0.0)) bad = lane + 1;
        sc[lane] = g > 0.0 ? 1.0 / sqrt(g) : 0.0;
    }
    wsync();
    for (int q = lane; q < d * d; q += 64) G[q / d][q % d] *= sc[q / d] * sc[q % d];
    for (int q = lane; q < d * n; q += 64) H[q / n][q % n] *= sc[q / n];
    wsync();
    for (int j = 0; j < d; ++j) {                        // right-looking Cholesky, forward substitution folded in
        double djj = G[j][j];
        if (!(djj > 1e-14)) {
            if (lane == 0 && bad == 0) bad = j + 1;
            djj = 1.0;
        }
        const double il = 1.0 / sqrt(djj);
        wsync();
        if (lane == j) G[j][j] = il;
        for (int r = j + 1 + lane; r < d; r += 64) G[r][j] *= il;
        for (int k = lane; k < n; k += 64) H[j][k] *= il;
        wsync();
        for (int q = lane; q < d * d; q += 64) {
            const int r = q / d, c = q % d;
            if (c > j && r >= c) G[r][c] -= G[r][j] * G[c][j];
        }
        for (int q = lane; q < d * n; q += 64) {
            const int r = q / n, k = q % n;
            if (r > j) H[r][k] -= G[r][j] * H[j][k];
        }
        wsync();
    }
    for (int i = d - 1; i >= 0; --i) {                   // L' w = y, column oriented
        const double il = G[i][i];
        wsync();
        for (int k = lane; k < n; k += 64) H[i][k] *= il;
        wsync();
        for (int q = lane; q < i * n; q += 64) H[q / n][q % n] -= G[i][q / n] * H[i][q % n];
    }
    wsync();
    for (int q = lane; q < n * n; q += 64) A[q] = H[q % n][q / n] * sc[q % n];                 // ABhat[k][i] = X[i][k]
    for (int q = lane; q < n * m; q += 64) B[q] = H[n + q % m][q / m] * sc[n + q % m];
    if (lane == 0) info[0] = bad;
}

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
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 2 && m == 1) hipLaunchKernelGGL((riccati_kernel_t<2, 1>), dim3(1), dim3(64), 0, st, a);
    else if (n == 12 && m == 4) hipLaunchKernelGGL((riccati_kernel_t<12, 4>), dim3(1), dim3(64), 0, st, a);
    else if (n == 5 && m == 2) hipLaunchKernelGGL((riccati_kernel_t<5, 2>), dim3(1), dim3(64), 0, st, a);
    else if (n == 6 && m == 2) hipLaunchKernelGGL((riccati_kernel_t<6, 2>), dim3(1), dim3(64), 0, st, a);
    else if (n == 7 && m == 4) hipLaunchKernelGGL((riccati_kernel_t<7, 4>), dim3(1), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(riccati_kernel, dim3(1), dim3(64), 0, st, a);
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

int irs_least_squares(int n, int m, int N, const double* dxdu, const double* deltaf, double* A, double* B,
                      int* info, void* stream) {
    IRS_CHECK_ARG(n > 0 && n <= kMaxN && m > 0 && m <= kMaxM && N > 0, "need 0<n<=32, 0<m<=16, N>0");
    IRS_CHECK_ARG(dxdu && deltaf && A && B && info, "null pointer");
    hipLaunchKernelGGL(lstsq_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), n, m, N, dxdu, deltaf,
                       A, B, info);
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

int irs_descent_run(const irs_descent_call* c, void* stream) {
    IRS_CHECK_ARG(c != nullptr, "null call struct");
    return irs_tvlqr_descent(c->model, c->params, c->n_params, c->T, c->At, c->Bt, c->ct, c->Q, c->Qd,
                             c->R, c->alpha_R, c->xd_trj, c->x0, c->K, c->k, c->x_new, c->u_new, c->cost,
                             c->info, stream);
}

int irs_tvlqr_descent(int model, const double* params, int n_params, int T, const double* At,
                      const double* Bt, const double* ct, const double* Q, const double* Qd,
                      const double* R, double alpha_R, const double* xd_trj, const double* x0,
                      double* K, double* k, double* x_new, double* u_new, double* cost, int* info,
                      void* stream) {
    return irs_tvlqr_descent_row(model, params, n_params, T, At, Bt, ct, Q, Qd, R, alpha_R, xd_trj, x0, K, k, x_new,
                                 u_new, cost, info, nullptr, nullptr, stream);
}

}  // extern "C"

// irs_tvlqr_descent + (row != null) the 8-int info row of the fused iterate, written by the same launch
int irs_tvlqr_descent_row(int model, const double* params, int n_params, int T, const double* At,
                          const double* Bt, const double* ct, const double* Q, const double* Qd,
                          const double* R, double alpha_R, const double* xd_trj, const double* x0,
                          double* K, double* k, double* x_new, double* u_new, double* cost, int* info,
                          const int* smooth_info, int* row, void* stream) {
    IRS_CHECK_ARG(T > 0 && At && Bt && ct && Q && Qd && R && xd_trj && x0 && K && k && x_new && u_new &&
                  cost && info, "bad argument");
    ModelParams p;
    int rc = irs_load_params(model, params, n_params, &p);
    if (rc != IRS_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    IRS_DISPATCH_MODEL(model, {
        RiccatiArgs a{At, Bt, ct, Q, Qd, R, xd_trj, K, k, info, alpha_R, Model::NX, Model::NU, T};
        hipLaunchKernelGGL((descent_kernel<Model>), dim3(1), dim3(64), 0, st, p, a, x0, x_new, u_new, cost, smooth_info,
                           row);
    });
    IRS_CHECK_LAUNCH();
    return IRS_OK;
}

#ifdef IRS_RIC_STAMPS
extern "C" void irs_debug_riccati_stamps(void) {
    long long h[8];
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ric_stamps), sizeof(h)) != hipSuccess) return;
    fprintf(stderr, "[riccati stamps] T=%lld  per step: head+14 MFMA %lld, H gather + LDL' %lld, column + substitutions + gain %lld, "
                    "10 MFMA tail %lld, waiting for the prefetched operands %lld cycles\n", h[4], h[0] / h[4], h[1] / h[4], h[2] / h[4], h[3] / h[4], h[5] / h[4]);
}
#endif
