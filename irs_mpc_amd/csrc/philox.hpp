// Philox4x32-10 counter-based generator (Salmon et al., SC'11) + Box-Muller.
// Specification restated in oracle/irs_oracle.py:device_gaussian_samples.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dual.hpp"

struct Philox4 { uint32_t v[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// Four N(0,1) draws for block j of sample `gidx` at timestep t, iteration it.
__device__ __forceinline__ void philox_normal4(uint64_t gidx, uint32_t t, uint32_t j, uint32_t it,
                                               uint64_t seed, float* z) {
    uint32_t c0 = (uint32_t)gidx;
    uint32_t c2 = j | ((uint32_t)(gidx >> 32) << 8);
    Philox4 r = philox4x32_10(c0, t, c2, it, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        // (r + 0.5) / 2^32 in (0,1); the f32 rounding of u keeps it inside (0,1]
        float u1 = ((float)r.v[2 * pr] + 0.5f) * 2.3283064365386963e-10f;
        float u2 = ((float)r.v[2 * pr + 1] + 0.5f) * 2.3283064365386963e-10f;
        // radius: hardware log2 / sqrt (1 ulp each); angle 2*pi*u2 reduced EXACTLY to a
        // quadrant (q/4 and u2 - q/4 are exact in f32), then the branch-free polynomials
        float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __log2f(u1));   // sqrt(-2 ln u1)
        float q = rintf(4.0f * u2);
        float r = (u2 - 0.25f * q) * 6.283185307179586f;
        float s, c;
        irs_sincos_quadrant(r, (int)q, s, c);
        z[2 * pr] = rad * c;
        z[2 * pr + 1] = rad * s;
    }
}
