// fuse.hip -- HBM-bound kernels of the LAFF hot path (gfx950): attention fusion over <= 8 feature planes,
// per-video frame attention, row normalisation + 16-bit operand packing.
//
// Reference arithmetic (all fp32; file:line under /root/reference):
//   Attention_1.forward                     model/Attention.py:78-105
//   Multi_head_MyApply_Attention.forward    model/Attention.py:508-531
//   no-transform branch (repeat + BN)       model/model.py:1801-1805, 1822-1823, 659-664, 1675-1676
//   frame attention loop                    model/model.py:2163-2173
//   loss.l2norm                             loss.py:8-13
//
// Mapping: one 64-lane wavefront per (row n, head h).  Lane i owns columns {256*j + 4*i .. +3} of the head
// (16-byte loads, 1 KiB contiguous per wave-instruction); the <= 8 softmax logits are reduced across the
// wave with xor-shuffles and the softmax itself is computed redundantly in every lane's registers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "exact_cos.h"
#include "wave_reduce.h"

namespace laff {

// all-lanes reductions of a wavefront: wave_reduce.h (the xor butterfly's order and bits, on lane swaps + DPP row rotations; as
// `__shfl_xor` loops fuse_reg_kernel<4, 2> held 101 ds_bpermute_b32 with their index arithmetic and LDS waits)
__device__ __forceinline__ float wave_sum(float v) { return wave_allsum(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_allmax(v); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scl4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 c) {
    return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// the same tanh / sigmoid as the GEMM epilogue (v_exp_f32 + v_rcp_f32), so that deferring the activation of a projection
// to this kernel leaves every bit of the result unchanged
__device__ __forceinline__ float plane_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.885390081777927f) + 1.0f);
}
__device__ __forceinline__ float plane_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// x_l[n,h,col..col+3] with the plane's activation, tiling and folded affine applied
__device__ __forceinline__ float4 load_plane(const FuseArgs& a, int l, long n, int h, int col) {
    const int srccol = a.tile[l] ? col : h * a.head_stride + col;
    typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
    // a projected plane is read exactly once: stream it past the L2 lines that hold the attention / BN vectors and Wt slices
    const nt_f32x4 nv = __builtin_nontemporal_load((const nt_f32x4*)(a.src[l] + n * a.ld[l] + srccol));
    float4 v = make_float4(nv.x, nv.y, nv.z, nv.w);
    const int act = a.act[l];                                   // wave-uniform
    if (act == LAFF_ACT_TANH) v = make_float4(plane_tanh(v.x), plane_tanh(v.y), plane_tanh(v.z), plane_tanh(v.w));
    else if (act == LAFF_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    else if (act == LAFF_ACT_SIGMOID) v = make_float4(plane_sigmoid(v.x), plane_sigmoid(v.y), plane_sigmoid(v.z), plane_sigmoid(v.w));
    if (a.scale[l]) {
        const int ai = a.tile[l] ? h * a.d + col : srccol;
        const float4 s = *(const float4*)(a.scale[l] + ai);
        const float4 t = *(const float4*)(a.shift[l] + ai);
        v = make_float4(fmaf(v.x, s.x, t.x), fmaf(v.y, s.y, t.y), fmaf(v.z, s.z, t.z), fmaf(v.w, s.w, t.w));
    }
    if (a.rownorm[l]) v = scl4(v, a.rownorm[l][n]);             // wave-uniform; l2norm(local_embs, dim=2) of the expert branch
    return v;
}

// load_plane in two steps, for callers that hold several planes in registers: FIRST every raw load (nothing between them but scalar
// tests: all of them are in flight together), THEN activation / folded affine / row norm.  As one call per (plane, chunk) hipcc had to
// finish each value -- the activation switch and the `scale` test are control flow on the loaded data's path -- before it could issue the
// next load: eight dependent HBM round trips per wavefront in fuse_reg_kernel<4, 2> (found in the ISA: `global_load ... nt ; s_waitcnt
// vmcnt(0)` eight times over), the 72 % of its cycles that wavefront spent waiting.  Bit for bit the same values as load_plane.
__device__ __forceinline__ float4 load_plane_raw(const FuseArgs& a, int l, long n, int h, int col) {
    const int srccol = a.tile[l] ? col : h * a.head_stride + col;
    typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
    const nt_f32x4 nv = __builtin_nontemporal_load((const nt_f32x4*)(a.src[l] + n * a.ld[l] + srccol));
    return make_float4(nv.x, nv.y, nv.z, nv.w);
}
template <int NCH>
__device__ __forceinline__ void finish_plane(const FuseArgs& a, int l, long n, int h, int lane, int d, float4 (&x)[NCH]) {
    const int act = a.act[l];                                   // wave-uniform
    if (act != LAFF_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            float4 v = x[j];
            if (act == LAFF_ACT_TANH) v = make_float4(plane_tanh(v.x), plane_tanh(v.y), plane_tanh(v.z), plane_tanh(v.w));
            else if (act == LAFF_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            else if (act == LAFF_ACT_SIGMOID) v = make_float4(plane_sigmoid(v.x), plane_sigmoid(v.y), plane_sigmoid(v.z), plane_sigmoid(v.w));
            x[j] = v;
        }
    }
    if (a.scale[l]) {
        float4 sc[NCH], sh[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int col = j * 256 + lane * 4;
            const int ai = col < d ? (a.tile[l] ? h * a.d + col : h * a.head_stride + col) : 0;    // (columns beyond d: any valid address)
            sc[j] = *(const float4*)(a.scale[l] + ai);
            sh[j] = *(const float4*)(a.shift[l] + ai);
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            x[j] = make_float4(fmaf(x[j].x, sc[j].x, sh[j].x), fmaf(x[j].y, sc[j].y, sh[j].y), fmaf(x[j].z, sc[j].z, sh[j].z),
                               fmaf(x[j].w, sc[j].w, sh[j].w));
    }
    if (a.rownorm[l]) {                                          // wave-uniform; l2norm(local_embs, dim=2) of the expert branch
        const float rn = a.rownorm[l][n];
#pragma unroll
        for (int j = 0; j < NCH; ++j) x[j] = scl4(x[j], rn);
    }
    // columns beyond d hold zeros, whatever the activation made of them
#pragma unroll
    for (int j = 0; j < NCH; ++j)
        if (j * 256 + lane * 4 >= d) x[j] = make_float4(0, 0, 0, 0);
}

// out[l][n] = 1 / (|x_l[n, :]|_2 + 1e-13 + 1e-14) over ALL columns of the stacked plane (every head): the factor that
// `local_embs = l2norm(local_embs, dim=2)` applies after the expert embedding was added (model/model.py:1866-1873, :1686-1694;
// loss.l2norm, loss.py:8-13).  One wavefront per (row, plane).
__global__ __launch_bounds__(256) void plane_row_norms_kernel(FuseArgs a, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)a.N * a.L) return;
    const long n = item / a.L;
    const int l = (int)(item - n * a.L);
    const int nh = a.head_stride ? a.H : 1;                      // without split heads the stacked row has d columns
    float ss = 0.f;
    for (int h = 0; h < nh; ++h)
        for (int col = lane * 4; col < a.d; col += 256) {
            const float4 v = load_plane(a, l, n, h, col);
            ss += dot4(v, v);
        }
    ss = wave_sum(ss);
    if (lane == 0) out[(long)l * a.N + n] = 1.0f / (sqrtf(ss) + 1e-13f + 1e-14f);
}

hipError_t launch_plane_row_norms(const FuseArgs& a, float* out, hipStream_t st) {
    const long items = (long)a.N * a.L;
    hipLaunchKernelGGL(plane_row_norms_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a, out);
    return hipGetLastError();
}

// gather plane (sparse feature through its FC, fc_gather_kernel's arithmetic inside the fuse launch): the projected plane is
// never written to / re-read from HBM.  The caption's ids arrive with one coalesced load per 64 and are broadcast with
// v_readlane; all NCH column chunks of the (row, head) are accumulated in the same pass over the ids.
template <int NCH>
__device__ __forceinline__ void load_gather_plane(const FuseArgs& a, int l, long n, int h, int lane, int d, float4 (&out)[NCH]) {
    const int beg = a.g_indptr[l][n], end = a.g_indptr[l][n + 1];
    const float* wt = a.g_wt[l];
    const long ldwt = a.g_ldwt[l];
    const int dk = a.g_dk[l];
    const int cbase = h * a.head_stride;                         // gather planes are never tiled
    float4 acc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc[j] = make_float4(0, 0, 0, 0);
    for (int p0 = beg; p0 < end; p0 += 64) {
        const int cnt = min(64, end - p0);
        int wi = 0;
        float vi = 0.0f;
        if (lane < cnt) {
            wi = a.g_indices[l][p0 + lane];
            vi = a.g_values[l] ? a.g_values[l][p0 + lane] : 1.0f;
            if (wi < 0 || wi >= dk) { wi = 0; vi = 0.0f; }
        }
        if (d == 256 * NCH) {
            // Whole heads (d = 256 NCH: every lane has all its columns): the rows of GW words are requested together and only then
            // added, in word order.  (As the generic loop below hipcc emitted ONE load at a time with `s_waitcnt vmcnt(0)` behind it --
            // the `col < d` test makes every load its own EXEC-masked block with the same destination registers --: 28 dependent
            // round trips per item at 14 words, the 0.03 ms per row per caption of the C1 launch.)
#ifndef LAFF_GATHER_GW
#define LAFF_GATHER_GW 4
#endif
            constexpr int GW = LAFF_GATHER_GW;
            for (int q = 0; q < cnt; q += GW) {                  // lanes >= cnt hold (row 0, weight 0)
                float4 r[GW][NCH];
                float v[GW];
#pragma unroll
                for (int u = 0; u < GW; ++u) {
                    const int w = __builtin_amdgcn_readlane(wi, q + u);
                    v[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vi), q + u));
                    const float* row = wt + (long)w * ldwt + cbase + lane * 4;
#pragma unroll
                    for (int j = 0; j < NCH; ++j) r[u][j] = *(const float4*)(row + j * 256);
                }
#pragma unroll
                for (int u = 0; u < GW; ++u)
#pragma unroll
                    for (int j = 0; j < NCH; ++j) acc[j] = fma4(r[u][j], v[u], acc[j]);
            }
            continue;
        }
        for (int q = 0; q < cnt; q += 4) {                       // lanes >= cnt hold (row 0, weight 0)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = __builtin_amdgcn_readlane(wi, q + u);
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vi), q + u));
                const float* row = wt + (long)w * ldwt + cbase;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int col = j * 256 + lane * 4;
                    if (col < d) acc[j] = fma4(*(const float4*)(row + col), v, acc[j]);
                }
            }
        }
    }
    const int act = a.act[l];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = j * 256 + lane * 4;
        float4 v = acc[j];
        if (col < d) {
            const int c = cbase + col;
            if (a.g_bias[l]) v = add4(v, *(const float4*)(a.g_bias[l] + c));
            if (act == LAFF_ACT_TANH) v = make_float4(plane_tanh(v.x), plane_tanh(v.y), plane_tanh(v.z), plane_tanh(v.w));
            else if (act == LAFF_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            else if (act == LAFF_ACT_SIGMOID) v = make_float4(plane_sigmoid(v.x), plane_sigmoid(v.y), plane_sigmoid(v.z), plane_sigmoid(v.w));
            if (a.scale[l]) {
                const float4 s = *(const float4*)(a.scale[l] + c);
                const float4 t = *(const float4*)(a.shift[l] + c);
                v = make_float4(fmaf(v.x, s.x, t.x), fmaf(v.y, s.y, t.y), fmaf(v.z, s.z, t.z), fmaf(v.w, s.w, t.w));
            }
        }
        out[j] = v;
    }
}

__device__ __forceinline__ void store16x4(void* base, long idx, float4 v, float scale, int bf16) {
    if (bf16) {
        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
        b4 o = {(__bf16)(v.x * scale), (__bf16)(v.y * scale), (__bf16)(v.z * scale), (__bf16)(v.w * scale)};
        *(b4*)((__bf16*)base + idx) = o;
    } else {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 o = {(_Float16)(v.x * scale), (_Float16)(v.y * scale), (_Float16)(v.z * scale), (_Float16)(v.w * scale)};
        *(h4*)((_Float16*)base + idx) = o;
    }
}

// softmax over L logits held identically by every lane
template <int L>
__device__ __forceinline__ void softmax_L(float (&lg)[L]) {
    float m = lg[0];
#pragma unroll
    for (int l = 1; l < L; ++l) m = fmaxf(m, lg[l]);
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        lg[l] = expf(lg[l] - m);
        s += lg[l];
    }
    const float inv = 1.0f / s;
#pragma unroll
    for (int l = 0; l < L; ++l) lg[l] *= inv;
}

// ---- laff_rank_prepare's work for the rows of a block, right behind their production (FuseArgs::rp_*; H == 1) -----------------------
//   band: q = |operand / prescale - e / (|e| + eps)|_2 of the 16-bit operand this launch emits for the row (fp64 sums),
//         band_t = q (1 + u) 1.0001 + c_acc,  band_v = q 1.0001  -- rank.hip has the derivation (rank_prepare_kernel does the same from memory);
//   text side: s_gt64 = exact(row, its ground-truth video) with exact_cos_with's arithmetic -- 16-lane groups walk the row exactly as
//         laff_rank_resolve walks a row in global memory: bit-equal scores for equal rows --, the count accumulator and the pair-list
//         header cleared, and (blocks [0, ceil(Nv / 64))) the maxima of the partner's band_v over its aligned 64-column blocks, which
//         the video-side launch in front of this one has completed per column.
template <int NCH>
__device__ __forceinline__ void rank_side(const FuseArgs& a, long n, int h, int lane, const float4 (&g)[NCH]) {
    __shared__ __attribute__((aligned(16))) float erow[4][256 * NCH];
    __shared__ long item_n[4];
    __shared__ int item_h[4];
    const int wave = threadIdx.x >> 6, d = a.d, H = a.H;
    // The block's four (row, head) items go through LDS and ONE wavefront does the fp64 work, a 16-lane group per item (as
    // rank_prepare_kernel and laff_rank_resolve walk rows): with every wave measuring and scoring its own item the launch grew by 46 us
    // at C4 -- more than the 39 us rank_prepare launch it replaces.
#pragma unroll
    for (int j = 0; j < NCH; ++j)
        if (j * 256 + lane * 4 < d) *(float4*)(&erow[wave][j * 256 + lane * 4]) = g[j];
    if (wave == 0 && lane < 4) item_n[lane] = -1;                   // (waves of a last, partial block that have no item: rank_side_idle)
    __syncthreads();
    if (lane == 0) { item_n[wave] = n; item_h[wave] = h; }
    __syncthreads();
    if (wave != 0) return;
    const int grp = lane >> 4, sl = lane & (RG - 1);
    const long nr = item_n[grp];
    const bool have = nr >= 0;
    const int hr = have ? item_h[grp] : 0;
    const long K = (long)H * d;
    const float* er = erow[grp];
    auto row = [&](int, int col) { return *(const float4*)(er + col); };
    double tt = 0.0, ch = 0.0;
    bool own = false;
    if (a.rp_side == 1) {
        const int c = have ? a.rp_gt[nr] - a.rp_col0 : -1;
        own = c >= 0 && c < a.rp_Nv;
        // one head's term of exact(): exact_cos_with on the head slice with H = 1 returns tv / ((|t_h| + eps)(|v_h| + eps)) unchanged.
        // (groups without a ground-truth row here walk row 0 of the partner: the shuffles of the reduction stay convergent)
        ch = exact_cos_with(row, a.rp_Ev + (long)(own ? c : 0) * K + (long)hr * d, 1, d, sl, &tt);
    } else {
        for (int col = sl * 4; col < d; col += RG * 4) {
            const float4 e = row(0, col);
            const double x = e.x, y = e.y, z = e.z, w = e.w;
            tt = fma(x, x, tt); tt = fma(y, y, tt); tt = fma(z, z, tt); tt = fma(w, w, tt);
        }
        tt = group_sum_f64(tt);
    }
    // q_h^2 = |operand / prescale - e / (|e| + eps)|^2 over the head, of the operand this launch has emitted
    const double inv_n = 1.0 / (sqrt(tt) + COS_EPS), inv_ps = 1.0 / (double)a.e16_scale;
    double q2 = 0.0;
    for (int col = sl * 4; col < d; col += RG * 4) {
        const float4 e4 = row(0, col);
        const float v4[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x16 = a.e16_bf16 ? (float)(__bf16)(v4[e] * a.e16_scale) : (float)(_Float16)(v4[e] * a.e16_scale);
            const double dlt = (double)x16 * inv_ps - (double)v4[e] * inv_n;
            q2 = fma(dlt, dlt, q2);
        }
    }
    q2 = group_sum_f64(q2);
    bool last = have;                                              // this group finishes the row (all of it with one head)
    if (H > 1) {
        // several heads: the per-head terms of a row come from up to H blocks.  Every item leaves {term, q_h^2} in the scratch and
        // takes a ticket; the H-th arrival sums the terms IN HEAD ORDER -- the sequence of exact_cos_with over the heads, so that
        // s_gt64 keeps the arithmetic of laff_rank_resolve whatever order the blocks ran in.
        // (agent-scope relaxed atomics, ordered by waiting for the stores' acknowledgement: a release / acquire fence pair here writes
        // back and invalidates the XCD's L2 once per item -- it made the launch 8 x longer)
        int tk = 0;
        if (have && sl == 0) {
            double* p = a.rp_part + ((size_t)nr * H + hr) * 2;
            __hip_atomic_store(p, ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p + 1, q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tk = (int)__hip_atomic_fetch_add(a.rp_ticket + nr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        tk = __shfl(tk, lane & ~(RG - 1));
        last = have && tk == H - 1;
        if (last) {
            double* p = a.rp_part + (size_t)nr * H * 2;
            double ssum = 0.0, qsum = 0.0;
            for (int hh = 0; hh < H; ++hh) {
                ssum += __hip_atomic_load(p + 2 * hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                qsum += __hip_atomic_load(p + 2 * hh + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ch = ssum / (double)H;
            q2 = qsum;
        }
    }
    if (last && sl == 0) {
        const float q = (float)sqrt(q2) * (1.0f / sqrtf((float)H));
        if (a.rp_side == 1) {
            a.rp_sgt[nr] = own ? ch : -INFINITY;
            a.rp_band[nr] = q * (1.0f + a.rp_unit) * 1.0001f + a.rp_cacc;
            a.rp_count[nr] = 0;
        } else {
            a.rp_band[nr] = q * 1.0001f;
        }
    }
    if (a.rp_side != 1) return;
    if (blockIdx.x == 0 && threadIdx.x < 4) a.rp_pairs[threadIdx.x] = 0u;
    const int nblk = (a.rp_Nv + 63) >> 6;
    if ((int)blockIdx.x < nblk) {
        const int v = (int)blockIdx.x * 64 + lane;
        float m = v < a.rp_Nv ? a.rp_band_v[v] : 0.0f;
        m = wave_max(m);
        if (lane == 0) a.rp_band_v[((a.rp_Nv + 3) & ~3) + blockIdx.x] = m;
    }
}

// a wavefront of a last, partial block that has no item: with rp_side set the block's other waves meet at rank_side's two barriers --
// it meets them there too instead of leaving (a barrier that counts on terminated waves being dropped is undefined in HIP)
__device__ __forceinline__ void rank_side_idle(const FuseArgs& a) {
    if (a.rp_side) {
        __syncthreads();
        __syncthreads();
    }
}

// ---- register-resident variant: d <= 256*NCH ----------------------------------------------------------------
// Six wavefronts per SIMD (at most 80 VGPRs; <4, 2> took 84 = five): the launch is a chain of load -> ~700 VALU -> store per wavefront that
// waits 72 % of its cycles (SQ_WAIT_ANY), so resident wavefronts are what keeps loads in flight.  C4 fuse launches 0.153 -> 0.142 ms,
// C5 2.15 -> 2.01 (A/B on one box, profiles/r6_fuse_occupancy.txt); eight per SIMD (64 VGPRs) spills and loses (0.181).  The step gains
// only a third of that: the pass is bound by its energy and the shorter launch draws what it saved.
// (More than eight resident float4 planes per lane -- L * NCH > 8 --, and L = 8, do not fit 80 registers: those instantiations stay unconstrained.)
#ifndef LAFF_FUSE_WAVES
#define LAFF_FUSE_WAVES 6
#endif
#ifndef LAFF_FUSE_WAVES8
#define LAFF_FUSE_WAVES8 5          // eight resident float4 planes per lane with all their loads in flight: 88 registers
#endif
#define LAFF_FUSE_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(L * NCH <= 6 ? LAFF_FUSE_WAVES : (L * NCH <= 8 && L < 8) ? LAFF_FUSE_WAVES8 : 1)))
template <int L, int NCH>
__global__ __launch_bounds__(256) LAFF_FUSE_WAVES_ATTR void fuse_reg_kernel(FuseArgs a) {
    const int lane = threadIdx.x & 63;
    long n;
    int h;
    if (a.head_major) {                                          // gather planes: all four waves of a block share the head
        h = (int)(blockIdx.x % (unsigned)a.H);
        n = (long)(blockIdx.x / (unsigned)a.H) * 4 + (threadIdx.x >> 6);
        if (n >= a.N) { rank_side_idle(a); return; }
    } else {
        const long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (it >= (long)a.N * a.H) { rank_side_idle(a); return; }
        if (a.H == 1) {                                          // (one head: no 64-bit division per wavefront)
            n = it;
            h = 0;
        } else {
            n = it / a.H;
            h = (int)(it - n * a.H);
        }
    }
    // one item per wavefront: (n, h) are wave-uniform, but derived from threadIdx they look divergent to the compiler -- pinned to
    // scalars, every row base (src + n * ld, E + item * d, ...) becomes SALU work and the loads take the scalar-base + 32-bit lane
    // offset form: the 64-bit address VALU (a tenth of this VALU-bound kernel's vector instructions) disappears
    n = ((long)__builtin_amdgcn_readfirstlane((int)(n >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)n);
    h = __builtin_amdgcn_readfirstlane(h);
    const long item = n * a.H + h;
    const int d = a.d;

    float4 x[L][NCH];
    float4 wv[NCH];
    // every dense plane's raw values first: L * NCH loads in flight together (see load_plane_raw).  The planes' kernel arguments are
    // read up front (as part of each conditional load they were one scalar load + wait per plane chunk, between the vector loads), and
    // whole heads (d = 256 NCH: every lane has all its columns) take loads without a per-lane test (no EXEC-masked block per load).
    const float* psrc[L];
    long pld[L];
    bool pdense[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        psrc[l] = a.src[l] + (a.tile[l] ? 0 : h * a.head_stride);
        pld[l] = a.ld[l];
        pdense[l] = !a.g_wt[l];
    }
    typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
    if (d == 256 * NCH) {
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                x[l][j] = make_float4(0, 0, 0, 0);
                if (pdense[l]) {                                             // wave-uniform
#ifdef LAFF_FUSE_PLANE_PLAIN
                    const nt_f32x4 nv = *(const nt_f32x4*)(psrc[l] + n * pld[l] + j * 256 + lane * 4);
#else
                    const nt_f32x4 nv = __builtin_nontemporal_load((const nt_f32x4*)(psrc[l] + n * pld[l] + j * 256 + lane * 4));
#endif
                    x[l][j] = make_float4(nv.x, nv.y, nv.z, nv.w);
                }
            }
    } else {
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int col = j * 256 + lane * 4;
                x[l][j] = (pdense[l] && col < d) ? load_plane_raw(a, l, n, h, col) : make_float4(0, 0, 0, 0);
            }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = j * 256 + lane * 4;
        wv[j] = (col < d && a.w) ? *(const float4*)(a.w + (long)h * d + col) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        if (a.g_wt[l]) load_gather_plane<NCH>(a, l, n, h, lane, d, x[l]);      // wave-uniform branch
        else finish_plane<NCH>(a, l, n, h, lane, d, x[l]);
    }
    if (a.flags & LAFF_ATT_L2NORM_EACH_HEAD) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) ss += dot4(x[l][j], x[l][j]);
            const float inv = 1.0f / (sqrtf(wave_sum(ss)) + 1e-13f + 1e-14f);
#pragma unroll
            for (int j = 0; j < NCH; ++j) x[l][j] = scl4(x[l][j], inv);
        }
    }
    float4 sum[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        sum[j] = x[0][j];
#pragma unroll
        for (int l = 1; l < L; ++l) sum[j] = add4(sum[j], x[l][j]);
    }
    float4 g[NCH];
    float lg[L];
    if (a.flags & LAFF_ATT_JUST_AVERAGE) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) g[j] = scl4(sum[j], 1.0f / L);
    } else {
        const bool mul = a.flags & LAFF_ATT_MUL;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float p = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const float4 c = mul ? mul4(x[l][j], scl4(sum[j], 1.0f / L)) : x[l][j];
                p += dot4(c, wv[j]);
            }
            lg[l] = wave_sum(p) + a.b[h];
        }
        softmax_L<L>(lg);
        const float ave = (a.flags & LAFF_ATT_WITH_AVE) ? a.gw[h] : 0.0f;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            g[j] = scl4(x[0][j], lg[0]);
#pragma unroll
            for (int l = 1; l < L; ++l) g[j] = fma4(x[l][j], lg[l], g[j]);
            g[j] = fma4(sum[j], ave, g[j]);
            ss += dot4(g[j], g[j]);
        }
        const float inv = 1.0f / (sqrtf(wave_sum(ss)) + 1e-14f);
#pragma unroll
        for (int j = 0; j < NCH; ++j) g[j] = scl4(g[j], inv);
        if (a.attn_w && lane < L) {
            float v = lg[0];
#pragma unroll
            for (int l = 1; l < L; ++l) v = (lane == l) ? lg[l] : v;
            a.attn_w[item * L + lane] = v;
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = j * 256 + lane * 4;
        if (col < d) {
            typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(nt_f32x4{g[j].x, g[j].y, g[j].z, g[j].w}, (nt_f32x4*)(a.E + item * d + col));
            if (a.E16) store16x4(a.E16, item * d + col, g[j], a.e16_scale, a.e16_bf16);
        }
    }
    if (a.rp_side) rank_side<NCH>(a, n, h, lane, g);               // kernel-uniform
}

// ---- streaming variant: any d % 4 == 0 (planes re-read from L2; used for d > 512) -----------------------------
template <int L>
__global__ __launch_bounds__(256) void fuse_stream_kernel(FuseArgs a) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)a.N * a.H) return;
    const long n = item / a.H;
    const int h = (int)(item - n * a.H);
    const int d = a.d;
    float invn[L];
#pragma unroll
    for (int l = 0; l < L; ++l) invn[l] = 1.0f;
    if (a.flags & LAFF_ATT_L2NORM_EACH_HEAD) {
        float ss[L];
#pragma unroll
        for (int l = 0; l < L; ++l) ss[l] = 0.f;
        for (int col = lane * 4; col < d; col += 256)
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const float4 v = load_plane(a, l, n, h, col);
                ss[l] += dot4(v, v);
            }
#pragma unroll
        for (int l = 0; l < L; ++l) invn[l] = 1.0f / (sqrtf(wave_sum(ss[l])) + 1e-13f + 1e-14f);
    }
    float lg[L];
    const bool javg = a.flags & LAFF_ATT_JUST_AVERAGE;
    if (!javg) {
        const bool mul = a.flags & LAFF_ATT_MUL;
        float p[L];
#pragma unroll
        for (int l = 0; l < L; ++l) p[l] = 0.f;
        for (int col = lane * 4; col < d; col += 256) {
            const float4 wv = *(const float4*)(a.w + (long)h * d + col);
            float4 xv[L];
            float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                xv[l] = scl4(load_plane(a, l, n, h, col), invn[l]);
                s = add4(s, xv[l]);
            }
            s = scl4(s, 1.0f / L);
#pragma unroll
            for (int l = 0; l < L; ++l) p[l] += dot4(mul ? mul4(xv[l], s) : xv[l], wv);
        }
#pragma unroll
        for (int l = 0; l < L; ++l) lg[l] = wave_sum(p[l]) + a.b[h];
        softmax_L<L>(lg);
        if (a.attn_w && lane < L) {
            float v = lg[0];
#pragma unroll
            for (int l = 1; l < L; ++l) v = (lane == l) ? lg[l] : v;
            a.attn_w[item * L + lane] = v;
        }
    }
    const float ave = (!javg && (a.flags & LAFF_ATT_WITH_AVE)) ? a.gw[h] : 0.0f;
    float ss = 0.f;
    for (int col = lane * 4; col < d; col += 256) {
        float4 g = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const float4 xv = scl4(load_plane(a, l, n, h, col), invn[l]);
            g = fma4(xv, javg ? 1.0f / L : lg[l] + ave, g);
        }
        ss += dot4(g, g);
        *(float4*)(a.E + item * d + col) = g;
    }
    if (javg) return;
    const float inv = 1.0f / (sqrtf(wave_sum(ss)) + 1e-14f);
    // each lane rescales exactly the elements it wrote itself
    for (int col = lane * 4; col < d; col += 256) {
        float4* e = (float4*)(a.E + item * d + col);
        const float4 v = scl4(*e, inv);
        *e = v;
        if (a.E16) store16x4(a.E16, item * d + col, v, a.e16_scale, a.e16_bf16);
    }
}

template <int L>
static hipError_t launch_fuse_L(const FuseArgs& a, hipStream_t st) {
    const long items = (long)a.N * a.H;
    const unsigned grid = a.head_major ? (unsigned)(((long)a.N + 3) / 4 * a.H) : (unsigned)((items + 3) / 4);
    if (a.d <= 256)
        hipLaunchKernelGGL((fuse_reg_kernel<L, 1>), dim3(grid), dim3(256), 0, st, a);
    else if (a.d <= 512)
        hipLaunchKernelGGL((fuse_reg_kernel<L, 2>), dim3(grid), dim3(256), 0, st, a);
    else if (a.rp_side)
        return hipErrorInvalidValue;           // laff_rank_prepare's work rides in the register-resident kernel only (d <= 512)
    else
        hipLaunchKernelGGL((fuse_stream_kernel<L>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_fuse(const FuseArgs& a, hipStream_t st) {
    switch (a.L) {
        case 1: return launch_fuse_L<1>(a, st);
        case 2: return launch_fuse_L<2>(a, st);
        case 3: return launch_fuse_L<3>(a, st);
        case 4: return launch_fuse_L<4>(a, st);
        case 5: return launch_fuse_L<5>(a, st);
        case 6: return launch_fuse_L<6>(a, st);
        case 7: return launch_fuse_L<7>(a, st);
        case 8: return launch_fuse_L<8>(a, st);
    }
    return hipErrorInvalidValue;
}

// ---- frame attention: one 256-thread workgroup per video ------------------------------------------------------
// Wave w walks frames w, w+4, ...; online softmax per wave, merged through LDS.  Columns: lane i owns
// {256*j + 4*i .. +3}, j < NCH (d <= 1024).
template <int NCH>
__global__ __launch_bounds__(256) void frame_fuse_kernel(FrameGroup grp) {
    __shared__ __attribute__((aligned(16))) float sh_acc[4][NCH * 256];
    __shared__ __attribute__((aligned(16))) float sh_sum[4][NCH * 256];
    __shared__ float sh_m[4], sh_s[4], sh_red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // all frame features of the tower share one launch: 3,000 videos alone are 1.5 rounds of workgroup slots, 4 x 3,000 fill them
    const int feat = (int)blockIdx.x / grp.f[0].B;
    const FrameArgs& a = grp.f[feat];
    const int bidx = (int)blockIdx.x - feat * a.B;
    const int d = a.d, Fmax = a.Fmax;
    int len = Fmax;
    if (a.mask) {
        // lens = mask_tensor.sum(dim = 1) (model/model.py:2156-2160 builds the row as ones followed by zeros): every wave sums the row
        float t = 0.f;
        for (int f = lane; f < Fmax; f += 64) t += a.mask[(long)bidx * a.ldm + f];
        len = min(max((int)(wave_sum(t) + 0.5f), 0), Fmax);
    } else if (a.lens) {
        len = min(max(a.lens[bidx], 0), Fmax);
    }
    const float* base = a.frames + (long)bidx * Fmax * d;
    const bool mul = a.flags & LAFF_ATT_MUL;
    const bool with_ave = a.flags & LAFF_ATT_WITH_AVE;
    const float bias = a.b[0];

    float4 wv[NCH], xs[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = j * 256 + lane * 4;
        wv[j] = col < d ? *(const float4*)(a.w + col) : make_float4(0, 0, 0, 0);
        xs[j] = make_float4(0, 0, 0, 0);
    }
    if (mul) {
        // mean over ALL Fmax frames (padded zeros included, as in the reference): w <- w * sum_f x_f / Fmax
        constexpr int FBM = NCH <= 2 ? 8 / NCH : 1;               // (frames requested together, added in frame order: see the main loop)
        for (int f0 = wave; f0 < len; f0 += 4 * FBM) {
            float4 xb[FBM][NCH];
#pragma unroll
            for (int i = 0; i < FBM; ++i) {
                const int f = min(f0 + 4 * i, len - 1);
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int col = j * 256 + lane * 4;
                    xb[i][j] = *(const float4*)(base + (long)f * d + (col < d ? col : 0));
                }
            }
#pragma unroll
            for (int i = 0; i < FBM; ++i) {
                if (f0 + 4 * i >= len) break;
#pragma unroll
                for (int j = 0; j < NCH; ++j)
                    if (j * 256 + lane * 4 < d) xs[j] = add4(xs[j], xb[i][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) *(float4*)&sh_sum[wave][j * 256 + lane * 4] = xs[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            float4 t = add4(add4(*(float4*)&sh_sum[0][c], *(float4*)&sh_sum[1][c]),
                            add4(*(float4*)&sh_sum[2][c], *(float4*)&sh_sum[3][c]));
            wv[j] = mul4(wv[j], scl4(t, 1.0f / Fmax));
            xs[j] = make_float4(0, 0, 0, 0);
        }
        __syncthreads();
    }
    float m = -INFINITY, s = 0.f;
    float4 acc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc[j] = make_float4(0, 0, 0, 0);
    // The wave's frames in batches of FB: every frame of a batch is requested before the first is used.  (One frame per trip, the
    // online softmax made each load wait for the previous frame's reduction and exponentials: eight dependent HBM round trips per
    // wave at 32 frames -- the launch was that chain, 0.10 ms at C3 for 0.4 GB.)  The frames are folded in the same order as before.
    constexpr int FB = NCH <= 2 ? 8 / NCH : 1;
    for (int f0 = wave; f0 < len; f0 += 4 * FB) {
        float4 xb[FB][NCH];
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            const int f = min(f0 + 4 * i, len - 1);                 // (frames beyond the clip: a valid address, never folded in)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int col = j * 256 + lane * 4;
                xb[i][j] = *(const float4*)(base + (long)f * d + (col < d ? col : 0));
            }
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            if (f0 + 4 * i >= len) break;                           // wave-uniform
            float4 xv[NCH];
            float p = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                xv[j] = (j * 256 + lane * 4 < d) ? xb[i][j] : make_float4(0, 0, 0, 0);
                p += dot4(xv[j], wv[j]);
            }
            const float lg = wave_sum(p) + bias;
            const float mn = fmaxf(m, lg);
            const float r = expf(m - mn), e = expf(lg - mn);
            s = s * r + e;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                acc[j] = fma4(xv[j], e, scl4(acc[j], r));
                xs[j] = add4(xs[j], xv[j]);
            }
            m = mn;
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        *(float4*)&sh_acc[wave][j * 256 + lane * 4] = acc[j];
        *(float4*)&sh_sum[wave][j * 256 + lane * 4] = xs[j];
    }
    if (lane == 0) {
        sh_m[wave] = m;
        sh_s[wave] = s;
    }
    __syncthreads();
    // merge the four partial softmaxes + the (Fmax - len) padded frames whose logit is exactly `bias`
    float M = fmaxf(fmaxf(sh_m[0], sh_m[1]), fmaxf(sh_m[2], sh_m[3]));
    const int npad = Fmax - len;
    if (npad > 0) M = fmaxf(M, bias);
    float S = npad > 0 ? npad * expf(bias - M) : 0.f;
    float sc[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        sc[w] = (sh_m[w] == -INFINITY) ? 0.f : expf(sh_m[w] - M);
        S += sh_s[w] * sc[w];
    }
    const float invS = 1.0f / S;
    const float ave = with_ave ? a.gw[0] : 0.f;
    // threads 0..d/4-1 of the block each finish one float4 of the output
    float ss = 0.f;
    float4 g = make_float4(0, 0, 0, 0);
    const int c4 = threadIdx.x * 4;
    constexpr int PER = 1;     // d <= 1024 columns = 256 threads x one float4
    float4 gq[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int c = q * 1024 + c4;
        gq[q] = make_float4(0, 0, 0, 0);
        if (c < d) {
            float4 t = make_float4(0, 0, 0, 0), u = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                t = fma4(*(float4*)&sh_acc[w][c], sc[w], t);
                u = add4(u, *(float4*)&sh_sum[w][c]);
            }
            g = fma4(u, ave, scl4(t, invS));
            gq[q] = g;
            ss += dot4(g, g);
        }
    }
    ss = wave_sum(ss);
    if (lane == 0) sh_red[wave] = ss;
    __syncthreads();
    const float inv = 1.0f / (sqrtf(sh_red[0] + sh_red[1] + sh_red[2] + sh_red[3]) + 1e-14f);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int c = q * 1024 + c4;
        if (c < d) *(float4*)(a.V + (long)bidx * d + c) = scl4(gq[q], inv);
    }
}

hipError_t launch_frame_fuse(const FrameGroup& g, hipStream_t st) {
    const FrameArgs& a = g.f[0];
    const dim3 grid((unsigned)((long)a.B * g.count));
    if (a.d <= 256)
        hipLaunchKernelGGL((frame_fuse_kernel<1>), grid, dim3(256), 0, st, g);
    else if (a.d <= 512)
        hipLaunchKernelGGL((frame_fuse_kernel<2>), grid, dim3(256), 0, st, g);
    else if (a.d <= 1024)
        hipLaunchKernelGGL((frame_fuse_kernel<4>), grid, dim3(256), 0, st, g);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- row normalisation + packing --------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ E, int N, int H, int d, int lde,
                                                        int normalize, float eps, float prescale, void* out) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)N * H) return;
    const long n = item / H;
    const int h = (int)(item - n * H);
    const float* src = E + n * lde + (long)h * d;
    const long K = (long)H * d;
    float mult = prescale;
    if constexpr (PREC == LAFF_PREC_FP32) {
        // generic (unaligned / d % 4 != 0) rows: scalar accesses, fp32 output only
        if ((d & 3) || (lde & 3) || ((uintptr_t)E & 15) || ((uintptr_t)out & 15)) {
            if (normalize) {
                float ss = 0.f;
                for (int col = lane; col < d; col += 64) ss += src[col] * src[col];
                mult = prescale / (sqrtf(wave_sum(ss)) + eps + 1e-14f);
            }
            for (int col = lane; col < d; col += 64) ((float*)out)[n * K + (long)h * d + col] = src[col] * mult;
            return;
        }
    }
    if (normalize) {
        float ss = 0.f;
        for (int col = lane * 4; col < d; col += 256) {
            const float4 v = *(const float4*)(src + col);
            ss += dot4(v, v);
        }
        mult = prescale / (sqrtf(wave_sum(ss)) + eps + 1e-14f);
    }
    for (int col = lane * 4; col < d; col += 256) {
        const float4 v = scl4(*(const float4*)(src + col), mult);
        const long o = n * K + (long)h * d + col;
        if constexpr (PREC == LAFF_PREC_FP32) {
            *(float4*)((float*)out + o) = v;
        } else if constexpr (PREC == LAFF_PREC_FP16 || PREC == LAFF_PREC_FP16X3) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            *(h4*)((_Float16*)out + o) = hi;
            if constexpr (PREC == LAFF_PREC_FP16X3) {
                h4 lo = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]),
                         (_Float16)(v.w - (float)hi[3])};
                *(h4*)((_Float16*)out + (long)N * K + o) = lo;
            }
        } else {
            typedef __bf16 b4 __attribute__((ext_vector_type(4)));
            b4 hi = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            *(b4*)((__bf16*)out + o) = hi;
            if constexpr (PREC == LAFF_PREC_BF16X3) {
                b4 lo = {(__bf16)(v.x - (float)hi[0]), (__bf16)(v.y - (float)hi[1]), (__bf16)(v.z - (float)hi[2]),
                         (__bf16)(v.w - (float)hi[3])};
                *(b4*)((__bf16*)out + (long)N * K + o) = lo;
            }
        }
    }
}

// ---- exact-product operand split for the FC GEMM on the 16-bit matrix pipe --------------------------------------
// x*s = hi + lo with s = 2^(9 - floor(log2 max|x_row|)) (row maximum lands in [512, 1024): no fp16 overflow for any input
// magnitude, lo stays normal down to 2^-34 of the row maximum), hi = fp16(x*s), lo = fp16(x*s - hi).  fp16 x fp16
// products are exact in fp32, so hi*hi' + hi*lo' + lo*hi' reproduces the fp32 product to ~2^-22 relative.
// out: [2][N][Kp] fp16 (hi plane, lo plane), columns >= K zero filled; rscale[n] = 1/s (a power of two: exact).
// binary exponent of a row maximum for the hi/lo split (row max * 2^(9-e) lands in [512, 1024)), straight from the bits: 0 for
// an all-zero / non-finite row, clamped to [-100, 127] so that both 2^(9-e) and 2^(e-9) stay normal floats (a row whose
// largest magnitude is below 2^-100 is all denormal after scaling either way)
__device__ __forceinline__ int split_exponent(float m) {
    const int be = (int)((__float_as_uint(m) >> 23) & 0xffu);
    if (!(m > 0.f) || be == 0xff) return 0;
    return max(be - 127, -100);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

struct SplitArgs {              // up to 8 matrices in one launch; block -> (matrix, group of 4 rows) by scalar search
    int count;
    int block_start[9];
    const float* X[8];
    int N[8], K[8], ldx[8], Kp[8];
    _Float16* out[8];
    float* rscale[8];
};

// NCH > 0: the row (K <= 256*NCH, 16-byte aligned) stays in registers between the max pass and the convert pass;
// NCH == 0: generic (any K / alignment), second pass re-reads the row through L2.
template <int NCH>
__global__ __launch_bounds__(256) void split_rows_kernel(SplitArgs g) {
    const int lane = threadIdx.x & 63;
    int p = 0;
    while (p + 1 < g.count && (int)blockIdx.x >= g.block_start[p + 1]) ++p;
    const long n = (long)((int)blockIdx.x - g.block_start[p]) * 4 + (threadIdx.x >> 6);
    const int N = g.N[p], K = g.K[p], ldx = g.ldx[p], Kp = g.Kp[p];
    if (n >= N) return;
    const float* src = g.X[p] + n * ldx;
    _Float16* hi = g.out[p] + n * Kp;
    _Float16* lo = g.out[p] + (long)N * Kp + n * Kp;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    float m = 0.f;
    if constexpr (NCH > 0) {
        float4 v[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            v[j] = c < K ? *(const float4*)(src + c) : make_float4(0, 0, 0, 0);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
        }
        m = wave_max(m);
        const int e = split_exponent(m);
        const float s = pow2f(9 - e);
        if (lane == 0) g.rscale[p][n] = pow2f(e - 9);
        if (!g.out[p]) return;                                   // scales only (laff_row_scales_grouped)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < Kp) {
                const float t[4] = {v[j].x * s, v[j].y * s, v[j].z * s, v[j].w * s};
                h4 h, l;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h[q] = (_Float16)t[q];
                    l[q] = (_Float16)(t[q] - (float)h[q]);
                }
                *(h4*)(hi + c) = h;
                *(h4*)(lo + c) = l;
            }
        }
    } else {
        for (int c = lane; c < K; c += 64) m = fmaxf(m, fabsf(src[c]));
        m = wave_max(m);
        const int e = split_exponent(m);
        const float s = pow2f(9 - e);
        if (lane == 0) g.rscale[p][n] = pow2f(e - 9);
        if (!g.out[p]) return;
        for (int c = lane * 4; c < Kp; c += 256) {       // Kp % 64 == 0, rows of the packed operand are 128-byte aligned
            h4 h, l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = (c + q < K ? src[c + q] : 0.f) * s;
                h[q] = (_Float16)t;
                l[q] = (_Float16)(t - (float)h[q]);
            }
            *(h4*)(hi + c) = h;
            *(h4*)(lo + c) = l;
        }
    }
}

hipError_t launch_split_rows_grouped(int count, const float* const* X, const int* N, const int* K, const int* ldx, void* const* out,
                                     float* const* rscale, hipStream_t st) {
    if (count < 1 || count > 8) return hipErrorInvalidValue;
    SplitArgs g{};
    g.count = count;
    int nb = 0, kmax = 0;
    bool vec = true;
    for (int i = 0; i < count; ++i) {
        g.block_start[i] = nb;
        nb += (N[i] + 3) / 4;
        g.X[i] = X[i]; g.N[i] = N[i]; g.K[i] = K[i]; g.ldx[i] = ldx[i]; g.Kp[i] = (K[i] + 63) / 64 * 64;
        g.out[i] = (_Float16*)out[i]; g.rscale[i] = rscale[i];
        kmax = K[i] > kmax ? K[i] : kmax;
        vec = vec && !(K[i] & 3) && !(ldx[i] & 3) && !((uintptr_t)X[i] & 15);
    }
    g.block_start[count] = nb;
    if (nb == 0) return hipSuccess;
    const dim3 grid((unsigned)nb), block(256);
    if (vec && kmax <= 256) hipLaunchKernelGGL((split_rows_kernel<1>), grid, block, 0, st, g);
    else if (vec && kmax <= 512) hipLaunchKernelGGL((split_rows_kernel<2>), grid, block, 0, st, g);
    else if (vec && kmax <= 1024) hipLaunchKernelGGL((split_rows_kernel<4>), grid, block, 0, st, g);
    else if (vec && kmax <= 2048) hipLaunchKernelGGL((split_rows_kernel<8>), grid, block, 0, st, g);
    else if (vec && kmax <= 4096) hipLaunchKernelGGL((split_rows_kernel<16>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((split_rows_kernel<0>), grid, block, 0, st, g);
    return hipGetLastError();
}

hipError_t launch_split_rows(const float* X, int N, int K, int ldx, int Kp, void* out, float* rscale, hipStream_t st) {
    (void)Kp;
    const float* xs[1] = {X};
    void* os[1] = {out};
    float* rs[1] = {rscale};
    return launch_split_rows_grouped(1, xs, &N, &K, &ldx, os, rs, st);
}

hipError_t launch_pack_rows(const float* E, int N, int H, int d, int lde, int normalize, float eps, float prescale,
                            int precision, void* out, hipStream_t st) {
    const long items = (long)N * H;
    const unsigned grid = (unsigned)((items + 3) / 4);
#define LAFF_PACK(P)                                                                                                \
    hipLaunchKernelGGL((pack_rows_kernel<P>), dim3(grid), dim3(256), 0, st, E, N, H, d, lde, normalize, eps, prescale, \
                       out)
    switch (precision) {
        case LAFF_PREC_FP32: LAFF_PACK(LAFF_PREC_FP32); break;
        case LAFF_PREC_FP16: LAFF_PACK(LAFF_PREC_FP16); break;
        case LAFF_PREC_BF16: LAFF_PACK(LAFF_PREC_BF16); break;
        case LAFF_PREC_FP16X3: LAFF_PACK(LAFF_PREC_FP16X3); break;
        case LAFF_PREC_BF16X3: LAFF_PACK(LAFF_PREC_BF16X3); break;
        default: return hipErrorInvalidValue;
    }
#undef LAFF_PACK
    return hipGetLastError();
}

}  // namespace laff

// ---- FC projection of a SPARSE feature ------------------------------------------------------
// The bag-of-words text feature (/root/reference/model/model.py:399-416, txt2vec.py) is a count vector over a vocabulary
// of thousands with ~10 non-zeros; `bow @ W^T` is then a weighted sum of ~10 columns of W.  One 256-thread workgroup per
// caption gathers those rows of W^T [Dk, D] (16-byte loads, D contiguous), accumulates in registers and applies the
// TransformNet epilogue (bias -> activation -> folded BN).  2*nnz*D flops instead of 2*Dk*D per row.
namespace laff {

__device__ __forceinline__ float gather_act(float v, int act) {
    if (act == 1) return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v * 2.885390081777927f) + 1.0f);
    if (act == 2) return fmaxf(v, 0.0f);
    if (act == 3) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
    return v;
}

// Work decomposition: the output is cut into column chunks of 512 floats (128 lanes x float4); workgroup b handles chunk
// b % nchunks of two captions.  Workgroups are dispatched round-robin over the 8 XCDs, so with nchunks == 8 (D = 4096)
// XCD x only ever touches columns [512x, 512x+512) of W^T: its private 4 MiB L2 then holds the ~2,000 most frequent
// vocabulary entries of that slice instead of the ~250 most frequent whole 16 KiB rows, and word frequencies are Zipfian.
__global__ __launch_bounds__(256) void fc_gather_kernel(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                        const float* __restrict__ values, int N, int Dk, const float* __restrict__ Wt,
                                                        long ldwt, const float* __restrict__ bias, const float* __restrict__ bn_scale,
                                                        const float* __restrict__ bn_shift, int D, int nchunks, int act,
                                                        float* __restrict__ Y, long ldy) {
    const int chunk = blockIdx.x % nchunks;
    const long n = (long)(blockIdx.x / nchunks) * 2 + (threadIdx.x >> 7);
    const int c_real = chunk * 512 + (threadIdx.x & 127) * 4;
    if (n >= N) return;                                        // wave-uniform (a wave never straddles two captions)
    const bool live = c_real < D;                              // lanes past D stay in the wave (they feed v_readlane) but
    const int c = live ? c_real : 0;                           // read column 0 and store nothing
    const int beg = indptr[n], end = indptr[n + 1];
    const int lane = threadIdx.x & 63;
    float4 acc0 = make_float4(0, 0, 0, 0), acc1 = acc0;
    const float* wcol = Wt + c;
    // The caption's ids/values are fetched 64 at a time with ONE coalesced load per wave and then broadcast lane by lane
    // (v_readlane): the row loads of W^T no longer wait on a dependent index load each, and 8 of them are in flight per lane.
    for (int p0 = beg; p0 < end; p0 += 64) {
        const int cnt = min(64, end - p0);
        int wi = 0;
        float vi = 0.0f;
        if (lane < cnt) {
            wi = indices[p0 + lane];
            vi = values ? values[p0 + lane] : 1.0f;
            if (wi < 0 || wi >= Dk) { wi = 0; vi = 0.0f; }     // ids outside the vocabulary contribute nothing
        }
        int j = 0;
        for (; j < cnt; j += 8) {                               // lanes >= cnt hold (row 0, weight 0): the tail needs no branch
            float4 r[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int w = __builtin_amdgcn_readlane(wi, j + u);
                v[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vi), j + u));
                r[u] = *(const float4*)(wcol + (long)w * ldwt);
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc0 = fma4(r[u], v[u], acc0);
                acc1 = fma4(r[u + 1], v[u + 1], acc1);
            }
        }
    }
    float o[4] = {acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = o[e] + (bias ? bias[c + e] : 0.0f);
        t = gather_act(t, act);
        if (bn_scale) t = fmaf(t, bn_scale[c + e], bn_shift[c + e]);
        o[e] = t;
    }
    typedef float gather_f32x4 __attribute__((ext_vector_type(4)));
    if (live) __builtin_nontemporal_store(gather_f32x4{o[0], o[1], o[2], o[3]}, (gather_f32x4*)(Y + n * ldy + c));
}

hipError_t launch_fc_gather(const int* indptr, const int* indices, const float* values, int N, int Dk, const float* Wt, int ldwt,
                            const float* bias, const float* bn_scale, const float* bn_shift, int D, int act, float* Y, int ldy,
                            hipStream_t st) {
    const int nchunks = (D + 511) / 512;
    const long blocks = (long)((N + 1) / 2) * nchunks;
    if (blocks > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fc_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, st, indptr, indices, values, N, Dk, Wt, (long)ldwt,
                       bias, bn_scale, bn_shift, D, nchunks, act, Y, (long)ldy);
    return hipGetLastError();
}

}  // namespace laff
