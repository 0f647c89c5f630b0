// exact_cos.h -- the exact score of the rank pipeline (rank.hip; also computed inside laff_fuse_packed_rank, fuse.hip):
//   exact(t, v) = (1/H) sum_h <t_h, v_h> / ((|t_h| + eps)(|v_h| + eps))     in fp64 on the fp32 embeddings
// (the infinitely precise value of what loss.cosine_sim computes in fp32, /root/reference/loss.py:8-13,30-34; products of fp32 values are
// exact in fp64).  Rows / pairs are handled by GROUPS OF 16 LANES: lane sl of a group owns the float4 columns {64 j + 4 sl}.
#pragma once
#include <hip/hip_runtime.h>

namespace laff {

constexpr int RG = 16;                         // lanes per row / pair
__device__ __forceinline__ double group_sum_f64(double v) {
#pragma unroll
    for (int o = RG / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float group_sum_f32(float v) {
#pragma unroll
    for (int o = RG / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

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
        // (unrolled: all 16 row loads of a d = 512 head in flight at once -- the resolve kernels are bound by the latency of these
        // scattered 2 KB rows, and as a rolled loop a pair was eight dependent round trips; the order of the fma chain is unchanged)
#pragma unroll 8
        for (int c = sl * 4; c < d; c += RG * 4) {
            const float4 a = load_t(h, c), b = *(const float4*)(vh + c);
#ifdef LAFF_EXACT_NOFMA
            tt += (double)(a.x + a.y + a.z + a.w); vv += (double)(b.x + b.y + b.z + b.w); tv += 1.0;
            continue;
#endif
#ifdef LAFF_EXACT_TVONLY                  /* timing only: what stored row norms would leave of the chains */
            { const double ax_ = a.x, ay_ = a.y, az_ = a.z, aw_ = a.w, bx_ = b.x, by_ = b.y, bz_ = b.z, bw_ = b.w;
              tv = fma(ax_, bx_, tv); tv = fma(ay_, by_, tv); tv = fma(az_, bz_, tv); tv = fma(aw_, bw_, tv); tt = 1.0; vv = 1.0; }
            continue;
#endif
            const double ax = a.x, ay = a.y, az = a.z, aw = a.w, bx = b.x, by = b.y, bz = b.z, bw = b.w;
            tt = fma(ax, ax, tt); tt = fma(ay, ay, tt); tt = fma(az, az, tt); tt = fma(aw, aw, tt);
            vv = fma(bx, bx, vv); vv = fma(by, by, vv); vv = fma(bz, bz, vv); vv = fma(bw, bw, vv);
            tv = fma(ax, bx, tv); tv = fma(ay, by, tv); tv = fma(az, bz, tv); tv = fma(aw, bw, tv);
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
