// Micro-benchmark (diagnostics): issue rate and dependent latency of v_mfma_f64_16x16x4_f64 on gfx950, one wave.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/microbench/mfma_f64_latency.hip -o /tmp/mf && /tmp/mf
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int N = 256;
__global__ __launch_bounds__(64) void k(double* out, long long* cyc) {
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0, t1;
    // (1) accumulate chain: D = A*B + D, same accumulator
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    asm volatile("" :: "v"(c0));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[0] = t1 - t0;
    // (2) four independent accumulators, round robin
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N / 4; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    asm volatile("" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[1] = t1 - t0;
    // (3) chain through the B operand: result register 0 feeds the next product
    t0 = __builtin_amdgcn_s_memtime();
    v4d z = {0, 0, 0, 0};
#pragma unroll 1
    for (int i = 0; i < N; ++i) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z, 0, 0, 0); b = c0[0] * 1e-3; }
    asm volatile("" :: "v"(c0));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[2] = t1 - t0;
    // (4) dependent f64 FMA chain (VALU) for comparison
    double x = a;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N; ++i) x = fma(x, b, a);
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[3] = t1 - t0;
    // (5) independent f64 FMAs (8 accumulators)
    double y[8] = {a, b, a, b, a, b, a, b};
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) y[q] = fma(y[q], b, a);
    }
    asm volatile("" :: "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[4] = t1 - t0;
    // (6) readlane broadcast chain: 24 v_readlane_b32 + 12 FMAs with SGPR operands (a 12 x 12 matvec row)
    double acc = a;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(acc), q), hi = __builtin_amdgcn_readlane(__double2hiint(acc), q);
            const double v = __hiloint2double(hi, lo);
            if (q & 1) s1 = fma(v, b, s1); else s0 = fma(v, a, s0);
        }
        acc = (s0 + s1) * 1e-3;
    }
    asm volatile("" :: "v"(acc));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[5] = t1 - t0;
    // (7) the forward step of the matrix-core descent: three chained products per step, the B operands are registers
    //     0..2 of the previous step's result (no VALU instruction in the chain)
    {
        v4d S = {a, b, a, b}, D;
        const double g0 = 1e-3 * a, g1 = 1e-3 * b, g2 = 2e-3 * a;
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int i = 0; i < N / 2; ++i) {
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(g0, S[0], z, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(g1, S[1], D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(g2, S[2], D, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f64_16x16x4f64(g0, D[0], z, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f64_16x16x4f64(g1, D[1], S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f64_16x16x4f64(g2, D[2], S, 0, 0, 0);
        }
        asm volatile("" :: "v"(S));
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[6] = t1 - t0;
        c1 += S;
    }
    // (8) the same with the A operands read from LDS one step ahead and two LDS writes per step by 4 lanes
    {
        __shared__ double gl[64 * 3 * 8];
        __shared__ double ol[N * 4 * 2];
        for (int q = lane; q < 64 * 3 * 8; q += 64) gl[q] = 1e-3 * (1 + (q & 7));
        __builtin_amdgcn_s_waitcnt(0);
        v4d S = {a, b, a, b}, D;
        double g0 = gl[lane], g1 = gl[64 + lane], g2 = gl[128 + lane], h0, h1, h2;
        const bool outl = (lane & 15) == 0;
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int i = 0; i < N / 2; ++i) {
            const int o = ((2 * i + 1) & 7) * 192 + lane, o2 = ((2 * i + 2) & 7) * 192 + lane;
            h0 = gl[o]; h1 = gl[o + 64]; h2 = gl[o + 128];
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(g0, S[0], z, 0, 0, 0);
            if (outl) { ol[(2 * i) * 8 + (lane >> 4)] = S[3]; ol[(2 * i) * 8 + 4 + (lane >> 4)] = S[3] > 0 ? 0.0 : S[3]; }
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(g1, S[1], D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(g2, S[2], D, 0, 0, 0);
            g0 = gl[o2]; g1 = gl[o2 + 64]; g2 = gl[o2 + 128];
            S = __builtin_amdgcn_mfma_f64_16x16x4f64(h0, D[0], z, 0, 0, 0);
            if (outl) { ol[(2 * i + 1) * 8 + (lane >> 4)] = D[3]; ol[(2 * i + 1) * 8 + 4 + (lane >> 4)] = D[3] > 0 ? 0.0 : D[3]; }
            S = __builtin_amdgcn_mfma_f64_16x16x4f64(h1, D[1], S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f64_16x16x4f64(h2, D[2], S, 0, 0, 0);
        }
        asm volatile("" :: "v"(S));
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[7] = t1 - t0;
        c2 += S;
        c3[0] += ol[lane];
    }
    out[lane] = c0[0] + c1[1] + c2[2] + c3[3] + x + y[0] + y[7] + acc + b;
}
int main() {
    double* out; long long* cyc; long long h[8];
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 8);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("mfma f64 16x16x4: accumulate chain %.1f cyc each; 4 independent %.1f cyc each; chain via B operand (incl. 1 mul) %.1f cyc\n",
           (double)h[0] / N, (double)h[1] / N, (double)h[2] / N);
    printf("v_fma_f64: dependent %.1f cyc; independent x8 %.1f cyc each; 12-term matvec row via readlane: %.1f cyc per row\n",
           (double)h[3] / N, (double)h[4] / N, (double)h[5] / (N / 8));
    printf("forward step of the descent (3 chained products, B = previous result): %.1f cyc per step; with its LDS reads and writes: %.1f\n",
           (double)h[6] / N, (double)h[7] / N);
    return 0;
}
