// strip_util.h -- small device helpers shared by the register-stationary ("strip") kernels (fc_strip.hip; sim_strip.hip keeps its
// own copies of the same idioms): compile-time loops, counted waits, raw buffer descriptors, values pinned in SGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

namespace laff {
namespace su {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// raw buffer descriptor of `bytes_total` bytes at `base`, re-based by `off` bytes (empty when off is beyond the end)
__device__ __forceinline__ u32x4 rebased_rsrc(unsigned long long base, unsigned long long bytes_total, unsigned long long off) {
    const unsigned long long b = base + off;
    const unsigned long long left = bytes_total > off ? bytes_total - off : 0ull;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xffffu;
    r.z = __builtin_amdgcn_readfirstlane((unsigned)(left > 0xffffffffull ? 0xffffffffull : left));
    r.w = 0x00020000u;
    return r;
}

__device__ __forceinline__ unsigned m0_get() {
    unsigned k;
    asm volatile("s_mov_b32 %0, m0" : "=s"(k)::"memory");
    return k;
}
__device__ __forceinline__ void m0_set(unsigned k) { asm volatile("s_mov_b32 m0, %0" ::"s"(k) : "memory"); }

// keep a wave-uniform value in an SGPR and opaque (no re-load from the kernarg segment inside hand-counted loops: scalar loads
// share lgkmcnt with the LDS reads and return out of order)
__device__ __forceinline__ unsigned pin_s(unsigned x) {
    x = __builtin_amdgcn_readfirstlane(x);
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ int pin_s(int x) { return (int)pin_s((unsigned)x); }
__device__ __forceinline__ float pin_s(float x) { return __uint_as_float(pin_s(__float_as_uint(x))); }
__device__ __forceinline__ unsigned long long pin_s(unsigned long long x) {
    return ((unsigned long long)pin_s((unsigned)(x >> 32)) << 32) | pin_s((unsigned)x);
}

// an MFMA's result is not interlocked against a reader that hipcc cannot see (asm): 8 passes of a 32x32x16 = 12+ wait states
__device__ __forceinline__ void mfma_drain_nops() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

}  // namespace su
}  // namespace laff
