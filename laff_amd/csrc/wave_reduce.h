// wave_reduce.h -- all-lanes reductions of a 64-lane wavefront in the order of the xor butterfly 32, 16, 8, 4, 2, 1, without the LDS
// crossbar.  `__shfl_xor` compiles to ds_bpermute_b32 (two for a 64-bit value) + five index instructions + s_waitcnt lgkmcnt(0) PER STEP.
// Here the xor-32 / xor-16 steps are v_permlane32_swap / v_permlane16_swap of the value with itself (gfx950: the two results are the lane's
// own value and its partner's -- for a commutative op every lane gets the butterfly step), the xor-8 .. xor-1 steps are rotations by
// 8, 4, 2, 1 inside the 16-lane DPP rows: after the xor-8 step the values have period 8 inside a row, then period 4, ..., so lane
// (i - r) mod 16 holds the very bits of lane i ^ r.  Every lane ends with bit for bit the butterfly's result (floating-point sums
// included: the association is the butterfly's).  ALL 64 LANES MUST BE ACTIVE at the call.
#pragma once
#include <hip/hip_runtime.h>

namespace laff {

template <int CTL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return __builtin_amdgcn_update_dpp(0, v, CTL, 0xf, 0xf, false); }

template <typename T>
struct WaveBits;
template <> struct WaveBits<float> {
    static __device__ __forceinline__ unsigned to(float v) { return __float_as_uint(v); }
    static __device__ __forceinline__ float from(unsigned u) { return __uint_as_float(u); }
};
template <> struct WaveBits<int> {
    static __device__ __forceinline__ unsigned to(int v) { return (unsigned)v; }
    static __device__ __forceinline__ int from(unsigned u) { return (int)u; }
};
template <> struct WaveBits<unsigned> {
    static __device__ __forceinline__ unsigned to(unsigned v) { return v; }
    static __device__ __forceinline__ unsigned from(unsigned u) { return u; }
};

// one butterfly step on a 32-bit value: STEP 32 / 16 (lane swaps) or a DPP row rotation (8, 4, 2, 1)
template <int STEP, typename T, typename Op>
__device__ __forceinline__ T wave_step32(T v, Op op) {
    const unsigned u = WaveBits<T>::to(v);
    if constexpr (STEP == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return op(WaveBits<T>::from(r[0]), WaveBits<T>::from(r[1]));
    } else if constexpr (STEP == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        return op(WaveBits<T>::from(r[0]), WaveBits<T>::from(r[1]));
    } else {
        return op(v, WaveBits<T>::from(dpp_u32<0x120 + STEP>(u)));          // row_ror:STEP
    }
}
// ... and on a 64-bit value (two halves travel, the op runs on the whole)
template <typename T>
__device__ __forceinline__ T bits_to64(unsigned long long u) { T t; __builtin_memcpy(&t, &u, 8); return t; }
template <int STEP, typename T, typename Op>
__device__ __forceinline__ T wave_step64(T v, Op op) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    if constexpr (STEP == 32 || STEP == 16) {
        unsigned a0, a1, b0, b1;
        if constexpr (STEP == 32) {
            const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            a0 = rl[0]; a1 = rl[1]; b0 = rh[0]; b1 = rh[1];
        } else {
            const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            a0 = rl[0]; a1 = rl[1]; b0 = rh[0]; b1 = rh[1];
        }
        return op(bits_to64<T>(((unsigned long long)b0 << 32) | a0), bits_to64<T>(((unsigned long long)b1 << 32) | a1));
    } else {
        const unsigned plo = dpp_u32<0x120 + STEP>(lo), phi = dpp_u32<0x120 + STEP>(hi);
        return op(v, bits_to64<T>(((unsigned long long)phi << 32) | plo));
    }
}
template <int STEP, typename T, typename Op>
__device__ __forceinline__ T wave_step(T v, Op op) {
    if constexpr (sizeof(T) == 8) return wave_step64<STEP>(v, op);
    else return wave_step32<STEP>(v, op);
}
// the whole butterfly, FIRST = 32 (a wavefront), 16 (the two 32-lane halves separately: lanes 0..31 / 32..63 each reduce among themselves)
// or 8 (every 16-lane DPP row by itself: the 16-lane groups of exact_cos.h)
template <int FIRST = 32, typename T, typename Op>
__device__ __forceinline__ T wave_allreduce(T v, Op op) {
    if constexpr (FIRST >= 32) v = wave_step<32>(v, op);
    if constexpr (FIRST >= 16) v = wave_step<16>(v, op);
    v = wave_step<8>(v, op);
    v = wave_step<4>(v, op);
    v = wave_step<2>(v, op);
    v = wave_step<1>(v, op);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_allsum(T v) { return wave_allreduce<32>(v, [](T a, T b) { return a + b; }); }
template <typename T>
__device__ __forceinline__ T wave_allmax(T v) { return wave_allreduce<32>(v, [](T a, T b) { return a > b ? a : b; }); }
template <typename T>
__device__ __forceinline__ T wave_allmin(T v) { return wave_allreduce<32>(v, [](T a, T b) { return a < b ? a : b; }); }

}  // namespace laff
