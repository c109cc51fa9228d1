#include <hip/hip_runtime.h>
__global__ void k(const float* __restrict__ src, float* out) {
    __shared__ float buf[64 * 4 * 2];
    const int lane = threadIdx.x;
    // each lane: 16 bytes from src + lane*4 floats -> LDS buf[lane*4 ...] (LDS address = base + lane*16 by hardware)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                     (__attribute__((address_space(3))) void*)buf, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += buf[lane * 4 + i] * (i + 1);
    out[lane] = s;
}
int main() {
    float h[256], *d, *o, ho[64];
    for (int i = 0; i < 256; ++i) h[i] = i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 9) { float e = 0; for (int i = 0; i < 4; ++i) e += (l * 4 + i) * (i + 1); printf("lane %d: %g (expect %g)\n", l, ho[l], e); }
    return 0;
}
