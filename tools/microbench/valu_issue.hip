// VALU issue rate on gfx950 by waves per SIMD: how many cycles one SIMD spends per wave64 v_fma_f32 when it holds 1, 2, 4
// waves (block = 256, 512, 1024 threads, one workgroup per CU), with register operands, with one scalar operand, and
// for v_pk_fma_f32.  Build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue ; run: ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void k(float* out, float s0, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    float sc = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s0)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    const float vs = s0 + threadIdx.x * 1e-9f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], vs, 1e-3f);
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(vs));
        } else if constexpr (KIND == 2) {
            const f2 v2 = f2{vs, vs};
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(v2), "v"(v2));
        } else {
            // a dependent chain: every instruction needs the previous one
#pragma unroll
            for (int i = 0; i < 16; ++i) a[0] = fmaf(a[0], vs, 1e-3f);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, float* out, int cus) {
    const int iters = 20000;
    for (int block : {64, 256, 512, 1024}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(block), 0, 0, out, 0.999f, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(block), 0, 0, out, 0.999f, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_wave = (double)iters * (KIND == 2 ? 8 : 16);
        const double waves_per_simd = block <= 256 ? (block >= 256 ? 1.0 : 0.25) : block / 256.0;
        // cycles of one SIMD per wave-instruction at 2.4 GHz (nominal)
        const double cyc = ms * 1e-3 * 2.4e9 / (inst_per_wave * (waves_per_simd < 1 ? 1 : waves_per_simd));
        printf("%-28s block %4d (%.2f waves/SIMD): %.3f ms -> %.2f nominal cycles per wave-instruction per SIMD\n", name,
               block, waves_per_simd, ms, cyc);
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out;
    hipMalloc(&out, (size_t)cus * 1024 * sizeof(float));
    run<0>("v_fma_f32 (VGPR operands)", out, cus);
    run<1>("v_fmac_f32 (one SGPR operand)", out, cus);
    run<2>("v_pk_fma_f32", out, cus);
    run<3>("v_fma_f32 dependent chain", out, cus);
    // one CU only: does the rate change when the rest of the chip is idle (clock / power)?
    run<0>("v_fma_f32, ONE workgroup", out, 1);
    return 0;
}
