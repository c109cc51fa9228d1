// Probes the operand / result layout of v_mfma_f64_16x16x4_f64 with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D) {   // A 16x16 (i,k), B 16x16 (k,j), row-major
    const int l = threadIdx.x, col = l & 15, rg = l >> 4;
    v4d acc = {0, 0, 0, 0};
    for (int s = 0; s < 4; ++s) {
        double a = A[col * 16 + (rg + 4 * s)];       // A[i=l&15][k=rg+4s]
        double b = B[(rg + 4 * s) * 16 + col];       // B[k=rg+4s][j=l&15]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(rg + 4 * r) * 16 + col] = acc[r];   // row=(l>>4)+4r, col=l&15
}
int main() {
    double hA[256], hB[256], hD[256], ref[256];
    for (int i = 0; i < 256; ++i) { hA[i] = (i * 7) % 13 - 6; hB[i] = (i * 5) % 11 - 5; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int q = 0; q < 16; ++q) s += hA[i * 16 + q] * hB[q * 16 + j]; ref[i * 16 + j] = s; }
    double *A, *B, *D; hipMalloc(&A, 2048); hipMalloc(&B, 2048); hipMalloc(&D, 2048);
    hipMemcpy(A, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(B, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D);
    hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { if (hD[i * 16 + j] != ref[i * 16 + j]) ++bad; if (hD[j * 16 + i] != ref[i * 16 + j]) ++badT; }
    printf("mismatches: as-assumed %d, transposed %d\n", bad, badT);
    for (int i = 0; i < 2; ++i) { for (int j = 0; j < 8; ++j) printf("%6.0f/%-6.0f", hD[i * 16 + j], ref[i * 16 + j]); printf("\n"); }
    return 0;
}
