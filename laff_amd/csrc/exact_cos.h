// exact_cos.h -- the exact score of the rank pipeline (rank.hip; also computed inside laff_fuse_packed_rank, fuse.hip):
//   exact(t, v) = (1/H) sum_h <t_h, v_h> / ((|t_h| + eps)(|v_h| + eps))     in fp64 on the fp32 embeddings
// (the infinitely precise value of what loss.cosine_sim computes in fp32, /root/reference/loss.py:8-13,30-34; products of fp32 values are
// exact in fp64).  Rows / pairs are handled by GROUPS OF 16 LANES: lane sl of a group owns the float4 columns {64 j + 4 sl}.
#pragma once
#include <hip/hip_runtime.h>

#include "wave_reduce.h"

namespace laff {

constexpr int RG = 16;                         // lanes per row / pair
// All-lanes sums of a 16-lane group (= one DPP row), xor butterfly 8, 4, 2, 1 as row rotations (wave_reduce.h: bit for bit what the
// `__shfl_xor` butterfly gave, without its ds_bpermute_b32 -- two per step for a double --, index arithmetic and LDS wait per step).
// All lanes of the row must be active (callers keep idle groups in step for that).
static_assert(RG == 16, "the group reductions below are DPP row operations");
__device__ __forceinline__ double group_sum_f64(double v) { return wave_allreduce<8>(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ float group_sum_f32(float v) { return wave_allreduce<8>(v, [](float a, float b) { return a + b; }); }

constexpr double COS_EPS = 1e-13 + 1e-14;      // loss.cosine_sim -> l2norm(eps=1e-13): X / (norm + eps + 1e-14)  (loss.py:8-13,30-34)

// Every lane of the group returns the same value; identical arithmetic (lane -> column map, fma order, reduction tree) wherever it
// is called, so equal rows give bit-equal scores: a duplicate of the ground-truth video ties with it exactly and is not counted.
// load_t(h, c): the float4 of the text row at head h, column c (a global row in rank.hip, the LDS copy of the row the fuse kernel has
// just produced in fuse.hip); v: the video row in global memory.
template <typename LoadT>
__device__ __forceinline__ double exact_cos_with(LoadT&& load_t, const float* __restrict__ v, int H, int d, int sl, double* tt_last = nullptr) {
    double s = 0.0;
    for (int h = 0; h < H; ++h) {
        const float* vh = v + (long)h * d;
        double tt = 0.0, vv = 0.0, tv = 0.0;
        auto chain = [&](const float4& a, const float4& b) {
#ifdef LAFF_EXACT_NOFMA
            tt += (double)(a.x + a.y + a.z + a.w); vv += (double)(b.x + b.y + b.z + b.w); tv += 1.0;
            return;
#endif
#ifdef LAFF_EXACT_TVONLY                  /* timing only: what stored row norms would leave of the chains */
            { const double ax_ = a.x, ay_ = a.y, az_ = a.z, aw_ = a.w, bx_ = b.x, by_ = b.y, bz_ = b.z, bw_ = b.w;
              tv = fma(ax_, bx_, tv); tv = fma(ay_, by_, tv); tv = fma(az_, bz_, tv); tv = fma(aw_, bw_, tv); tt = 1.0; vv = 1.0; }
            return;
#endif
            const double ax = a.x, ay = a.y, az = a.z, aw = a.w, bx = b.x, by = b.y, bz = b.z, bw = b.w;
            tt = fma(ax, ax, tt); tt = fma(ay, ay, tt); tt = fma(az, az, tt); tt = fma(aw, aw, tt);
            vv = fma(bx, bx, vv); vv = fma(by, by, vv); vv = fma(bz, bz, vv); vv = fma(bw, bw, vv);
            tv = fma(ax, bx, tv); tv = fma(ay, by, tv); tv = fma(az, bz, tv); tv = fma(aw, bw, tv);
        };
#ifndef LAFF_EXACT_CH
#define LAFF_EXACT_CH 4
#endif
        constexpr int CH = LAFF_EXACT_CH;                            // float4 columns of a lane per batch: 4 x 64 = 256 columns (8 loads in flight)
        if (d % (RG * 4 * CH) == 0) {     // (CH = 8 -- a whole 512-d head at once -- needs 126 registers in the resolve kernel: 4 wavefronts per SIMD, slower)
            // Heads of whole 256-column batches: the 8 row loads of a batch are ALL requested before the first product (arrays
            // filled first, chains afterwards, in column order: the fma sequence -- and every bit of the result -- is that of the
            // rolled loop below).  With `#pragma unroll 8` on that loop hipcc issued the loads in pairs with `s_waitcnt vmcnt(0)`
            // between them -- eight dependent round trips per pair of scattered 2 KB rows (found in the ISA), which is what the
            // resolve kernels were bound by.
            for (int c0 = sl * 4; c0 < d; c0 += RG * 4 * CH) {
                float4 a[CH], b[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    a[i] = load_t(h, c0 + i * (RG * 4));
                    b[i] = *(const float4*)(vh + c0 + i * (RG * 4));
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) chain(a[i], b[i]);
            }
        } else {
            for (int c = sl * 4; c < d; c += RG * 4) chain(load_t(h, c), *(const float4*)(vh + c));
        }
        tt = group_sum_f64(tt); vv = group_sum_f64(vv); tv = group_sum_f64(tv);
        if (tt_last) *tt_last = tt;                                  // (|t_h|^2 of the last head, for callers that need the norm too)
        s += tv / ((sqrt(tt) + COS_EPS) * (sqrt(vv) + COS_EPS));
    }
    return s / (double)H;
}

__device__ __forceinline__ double exact_cos(const float* __restrict__ t, const float* __restrict__ v, int H, int d, int sl) {
    return exact_cos_with([&](int h, int c) { return *(const float4*)(t + (long)h * d + c); }, v, H, d, sl);
}

}  // namespace laff
