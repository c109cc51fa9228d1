#include <hip/hip_runtime.h>
__device__ __forceinline__ double row_bcast_d(double v, int) { return v; }
template <int N>
__device__ __forceinline__ double row_newbcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
    double v = threadIdx.x * 1.5;
    out[threadIdx.x] = row_newbcast<12>(v) + row_newbcast<13>(v) * 2.0;
}
int main() {
    double* d; hipMalloc(&d, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i += 7) printf("%d: %g (expect %g)\n", i, h[i], ((i / 16) * 16 + 12) * 1.5 + ((i / 16) * 16 + 13) * 3.0);
    return 0;
}
