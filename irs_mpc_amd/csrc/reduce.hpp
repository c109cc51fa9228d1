// Wave64 / workgroup reductions for arrays of per-lane accumulators (gfx950).
#pragma once
#include <hip/hip_runtime.h>

// Length the accumulator array is padded to so that the transpose-reduce below
// can halve it: next power of two for short arrays, next multiple of 64 otherwise.
constexpr int irs_reduce_pad(int P) {
    if (P > 32) return (P + 63) / 64 * 64;
    int q = 1;
    while (q < P) q <<= 1;
    return q;
}

// Transpose-reduce of LEN per-lane accumulators over the 64 lanes of a wave.
// Stage s pairs lanes that differ in bit (5-s).  While the live length is even the
// two partners split it (each keeps one half and receives the partner's copy of
// that half): ~LEN shuffles in total instead of 6*LEN.  Once the length is odd the
// remaining stages are plain butterflies.  On return lane l holds, in
// v[0..final_len), the wave totals of original indices base(l)+i, where
//   base(l) = sum over halving stages s of bit_(5-s)(l) * (len_s / 2);
// lanes that differ only in butterfly-stage bits hold identical copies.
template <int LEN, int STAGE>
struct WaveReduce {
    static constexpr bool HALVE = (LEN % 2 == 0);
    static constexpr int NEXT = HALVE ? LEN / 2 : LEN;
    static constexpr int FINAL_LEN = WaveReduce<NEXT, STAGE + 1>::FINAL_LEN;

    __device__ __forceinline__ static void run(float* v, int lane) {
        constexpr int mask = 32 >> STAGE;
        if constexpr (HALVE) {
            const bool upper = (lane & mask) != 0;
#pragma unroll
            for (int i = 0; i < LEN / 2; ++i) {
                float keep = upper ? v[i + LEN / 2] : v[i];
                float send = upper ? v[i] : v[i + LEN / 2];
                v[i] = keep + __shfl_xor(send, mask, 64);
            }
        } else {
#pragma unroll
            for (int i = 0; i < LEN; ++i) v[i] += __shfl_xor(v[i], mask, 64);
        }
        WaveReduce<NEXT, STAGE + 1>::run(v, lane);
    }
    // first original index owned by `lane`, or -1 if the lane holds a duplicate
    __device__ __forceinline__ static int base(int lane) {
        constexpr int mask = 32 >> STAGE;
        int rest = WaveReduce<NEXT, STAGE + 1>::base(lane);
        if constexpr (HALVE) {
            return (rest < 0) ? -1 : rest + ((lane & mask) ? LEN / 2 : 0);
        } else {
            return (lane & mask) ? -1 : rest;
        }
    }
};
template <int LEN>
struct WaveReduce<LEN, 6> {
    static constexpr int FINAL_LEN = LEN;
    __device__ __forceinline__ static void run(float*, int) {}
    __device__ __forceinline__ static int base(int) { return 0; }
};

// Reduces acc[0..P) (padded to PP = irs_reduce_pad(P) with zeros) over each wave and
// leaves the per-wave totals in LDS: red[w * PP + p], w < NW.  The caller barriers and
// adds the NW rows in a fixed order (deterministic run to run).
template <int P, int NW>
__device__ __forceinline__ void block_reduce_lds(float* acc, float* red) {
    constexpr int PP = irs_reduce_pad(P);
    using WR = WaveReduce<PP, 0>;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    WR::run(acc, lane);
    const int b = WR::base(lane);
    if (b >= 0) {
#pragma unroll
        for (int i = 0; i < WR::FINAL_LEN; ++i) red[wave * PP + b + i] = acc[i];
    }
}
