// rank.hip -- ranking kernels (gfx950): the argsort + label-matrix loop of /root/reference/predictor.py:232-244
// (text->video) and :262-270 (video->text) restated as counts, so no O(Nt*Nv) index or label matrix exists:
//   position(t, g) = 1 + #{ c != g : S[t,c] > S[t,g] }.
// HBM-bound: S is read once, 16 bytes per lane.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "kernels.h"
#include "exact_cos.h"
#include "wave_reduce.h"

// Cross-workgroup hand-off of the metrics tails (rank_metrics_kernel, rank_resolve_kernel).  Default: no fences -- the partials travel
// as device-scope stores / atomics, are complete once `s_waitcnt vmcnt(0)` returns, and the ticket is a relaxed device-scope atomic behind
// that wait: this is gfx9 hardware behaviour (vmcnt covers stores, device-scope atomics are performed memory-side), not a guarantee of
// the HIP memory model.  -DLAFF_TAIL_FENCES restores the model's own release / acquire pair around the tickets (an L2 write-back +
// invalidate per workgroup: 12 us slower at C4); tests/test_gpu_kernels.py::test_fence_free_tail_equals_the_fenced_build holds the two
// builds against each other over many replays.
#ifdef LAFF_TAIL_FENCES
#define LAFF_TICKET_ORDER __ATOMIC_ACQ_REL
#define LAFF_TAIL_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define LAFF_TAIL_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define LAFF_TICKET_ORDER __ATOMIC_RELAXED
#define LAFF_TAIL_RELEASE() do {} while (0)
#define LAFF_TAIL_ACQUIRE() do {} while (0)
#endif

namespace laff {

__global__ void gather_gt_kernel(const float* __restrict__ S, int Nt, int Nv, long lds, const int* __restrict__ gt_col,
                                 int col0, float* __restrict__ s_gt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Nt) return;
    const int c = gt_col[t] - col0;
    s_gt[t] = (c >= 0 && c < Nv) ? S[(long)t * lds + c] : -INFINITY;
}

// one 256-thread workgroup per text row
__global__ __launch_bounds__(256) void rank_count_kernel(const float* __restrict__ S, int Nt, int Nv, long lds,
                                                         const int* __restrict__ gt_col, int col0,
                                                         const float* __restrict__ s_gt, int* __restrict__ count,
                                                         int accumulate) {
    __shared__ int red[4];
    const int t = blockIdx.x;
    const float* row = S + (long)t * lds;
    const float sg = s_gt[t];
    const int gt = gt_col[t] - col0;
    int cnt = 0;
    const bool vec = ((lds & 3) == 0) && ((((uintptr_t)S) & 15) == 0);
    if (vec) {
        const int n4 = Nv >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float4 v = *(const float4*)(row + 4 * i);
            const int c = 4 * i;
            cnt += (v.x > sg && c != gt) + (v.y > sg && c + 1 != gt) + (v.z > sg && c + 2 != gt) + (v.w > sg && c + 3 != gt);
        }
        for (int c = (n4 << 2) + threadIdx.x; c < Nv; c += 256) cnt += (row[c] > sg && c != gt);
    } else {
        for (int c = threadIdx.x; c < Nv; c += 256) cnt += (row[c] > sg && c != gt);
    }
    cnt = wave_allsum(cnt);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = red[0] + red[1] + red[2] + red[3];
        if (accumulate) atomicAdd(count + t, tot); else count[t] = tot;
    }
}

// Video->text: a workgroup owns CT = 32 columns (videos) and streams all Nt rows; thread (ry, cx) compares its
// column's values against that column's <= G ground-truth thresholds held in LDS.
template <int G>
__global__ __launch_bounds__(256) void v2t_count_kernel(const float* __restrict__ S, int Nt, int Nv, long lds,
                                                        const int* __restrict__ grp_off, const int* __restrict__ grp_idx,
                                                        int* __restrict__ count, int pass) {
    constexpr int CT = 32;
    __shared__ float thr[CT][G + 1];
    __shared__ int cnts[8][CT][G + 1];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * CT + cx;
    const bool col_ok = c < Nv;
    int g0 = 0, gn = 0;
    if (col_ok) {
        g0 = grp_off[c] + pass * G;
        gn = min(max(grp_off[c + 1] - g0, 0), G);
    }
    if (ry == 0) {
        for (int i = 0; i < G; ++i)
            thr[cx][i] = (i < gn) ? S[(long)grp_idx[g0 + i] * lds + c] : INFINITY;
    }
    __syncthreads();
    float th[G];
    int cn[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        th[i] = thr[cx][i];
        cn[i] = 0;
    }
    if (col_ok) {
        for (int r = ry; r < Nt; r += 8) {
            const float v = S[(long)r * lds + c];
#pragma unroll
            for (int i = 0; i < G; ++i) cn[i] += (v > th[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) cnts[ry][cx][i] = cn[i];
    __syncthreads();
    if (ry == 0 && col_ok) {
        for (int i = 0; i < gn; ++i) {
            int tot = 0;
#pragma unroll
            for (int y = 0; y < 8; ++y) tot += cnts[y][cx][i];
            count[grp_idx[g0 + i]] = tot;
        }
    }
}

// s_gt[t] = scale * <T[t,:], V[gt[t]-col0,:]> on the packed 16-bit GEMM operands (fp32 accumulate, sequential per lane
// + wave tree), -inf when the column is not in this shard.  One wavefront per text row.  x3: hi*hi + hi*lo + lo*hi.
template <bool BF16>
__global__ __launch_bounds__(256) void row_dot_gt_kernel(const uint16_t* __restrict__ T, const uint16_t* __restrict__ V, int Nt,
                                                         int Nv, int K, int x3, float scale, const int* __restrict__ gt_col,
                                                         int col0, float* __restrict__ s_gt, int* __restrict__ zero_count) {
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= Nt) return;
    if (zero_count && lane == 0) zero_count[t] = 0;              // the fused count of laff_sim_gemm accumulates into this
    const int c = gt_col[t] - col0;
    if (c < 0 || c >= Nv) {
        if (lane == 0) s_gt[t] = -INFINITY;
        return;
    }
    auto cvt = [](uint16_t h) -> float {
        if constexpr (BF16) return __uint_as_float((unsigned)h << 16);
        else { _Float16 f; __builtin_memcpy(&f, &h, 2); return (float)f; }
    };
    const uint16_t* tr = T + t * K;
    const uint16_t* vr = V + (long)c * K;
    const long pT = (long)Nt * K, pV = (long)Nv * K;
    float acc = 0.f;
    if (K & 7) {                                   // rows are not 16-byte multiples: element-wise
        for (int k = lane; k < K; k += 64) {
            if (x3) acc += cvt(tr[pT + k]) * cvt(vr[k]) + cvt(tr[k]) * cvt(vr[pV + k]);
            acc = fmaf(cvt(tr[k]), cvt(vr[k]), acc);
        }
    } else
    for (int k = lane * 8; k < K; k += 512) {
        uint4 a = *(const uint4*)(tr + k), b = *(const uint4*)(vr + k);
        const uint16_t* pa = (const uint16_t*)&a;
        const uint16_t* pb = (const uint16_t*)&b;
        if (x3) {
            uint4 al = *(const uint4*)(tr + pT + k), bl = *(const uint4*)(vr + pV + k);
            const uint16_t* pal = (const uint16_t*)&al;
            const uint16_t* pbl = (const uint16_t*)&bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += cvt(pal[e]) * cvt(pb[e]) + cvt(pa[e]) * cvt(pbl[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(cvt(pa[e]), cvt(pb[e]), acc);
    }
    acc = wave_allsum(acc);
    if (lane == 0) s_gt[t] = acc * scale;
}

hipError_t launch_row_dot_gt(const void* T, const void* V, int Nt, int Nv, int K, int bf16, int x3, float scale,
                             const int* gt_col, int col0, float* s_gt, int* zero_count, hipStream_t st) {
    const unsigned grid = (unsigned)((Nt + 3) / 4);
    if (bf16)
        hipLaunchKernelGGL((row_dot_gt_kernel<true>), dim3(grid), dim3(256), 0, st, (const uint16_t*)T, (const uint16_t*)V, Nt, Nv, K, x3, scale, gt_col, col0, s_gt, zero_count);
    else
        hipLaunchKernelGGL((row_dot_gt_kernel<false>), dim3(grid), dim3(256), 0, st, (const uint16_t*)T, (const uint16_t*)V, Nt, Nv, K, x3, scale, gt_col, col0, s_gt, zero_count);
    return hipGetLastError();
}

// ---- exact ranks on a reduced-precision similarity GEMM ----------------------------------------------------------------
// The reference ranks on fp32 cosine scores (predictor.py:232-244 on model/model.py:1003-1016); a 16-bit MFMA pass is inside
// the 1e-4 score contract but moves ~3 % of the ranks by one place.  Ranks are index work, so they are made exact:
//   exact(t, v) = (1/H) sum_h <t_h, v_h> / ((|t_h| + eps)(|v_h| + eps))     in fp64 on the fp32 embeddings
// (the infinitely precise value of what loss.cosine_sim computes in fp32; products of fp32 values are exact in fp64).
//   * laff_rank_prepare : s_gt64[t] = exact(t, gt(t)); for every operand row the MEASURED quantisation error
//                         q = |operand/prescale - normalised embedding|_2, turned into band halves
//                             band_t[t] = q_t / sqrt(H) * (1 + u) + K * 2^-23 + 2^-20,   band_v[v] = q_v / sqrt(H)
//                         so that |approx(t,v) - exact(t,v)| <= band_t[t] + band_v[v]   (Cauchy-Schwarz on
//                         dt.v + t.dv + dt.dv with |t^| = |v^| = sqrt(H); u = unit roundoff of the operand format bounds the
//                         cross term; K * 2^-23 covers the fp32 accumulation of K exact products under round-to-nearest or
//                         truncation, 2^-20 the final scaling, the fp32 copy of s_gt64 and the rounding of the epilogue's thresholds);
//   * the GEMM epilogue decides every pair outside the band and lists the pairs inside it (gemm_nt.hip);
//   * laff_rank_resolve  re-scores the listed pairs with exact() and fixes count / S.
// Rows / pairs are handled by GROUPS OF 16 LANES (4 per wavefront): lane sl of a group owns the float4 columns {64 j + 4 sl}.
// Both kernels are latency-bound (a few KB per row, then a reduction), so 8 independent 16-byte loads per lane and a 4-step
// reduction beat one wavefront per row; 95k listed pairs at C4 are 24k wavefronts instead of 95k.
// 4 operand values of row `row` at column k (hi + lo for a split operand), PREC as LAFF_PREC_*
template <int PREC>
__device__ __forceinline__ void load_operand4(const void* __restrict__ op, long k, long plane, float (&x)[4]) {
    if constexpr (PREC == LAFF_PREC_FP32) {
        const float4 o = *(const float4*)((const float*)op + k);
        x[0] = o.x; x[1] = o.y; x[2] = o.z; x[3] = o.w;
    } else {
        auto cvt = [](uint16_t b) -> float {
            if constexpr (PREC == LAFF_PREC_BF16 || PREC == LAFF_PREC_BF16X3) return __uint_as_float((unsigned)b << 16);
            else { _Float16 f; __builtin_memcpy(&f, &b, 2); return (float)f; }
        };
        const uint2 o = *(const uint2*)((const uint16_t*)op + k);
        x[0] = cvt((uint16_t)o.x); x[1] = cvt((uint16_t)(o.x >> 16)); x[2] = cvt((uint16_t)o.y); x[3] = cvt((uint16_t)(o.y >> 16));
        if constexpr (PREC == LAFF_PREC_FP16X3 || PREC == LAFF_PREC_BF16X3) {
            const uint2 l = *(const uint2*)((const uint16_t*)op + plane + k);
            x[0] += cvt((uint16_t)l.x); x[1] += cvt((uint16_t)(l.x >> 16)); x[2] += cvt((uint16_t)l.y); x[3] += cvt((uint16_t)(l.y >> 16));
        }
    }
}

// load_operand4 in two steps (raw words first, conversion later), for callers that request several columns before using any
struct OperandRaw { uint2 hi, lo; float4 f; };
template <int PREC>
__device__ __forceinline__ void load_operand_raw(const void* __restrict__ op, long k, long plane, OperandRaw& r) {
    if constexpr (PREC == LAFF_PREC_FP32) {
        r.f = *(const float4*)((const float*)op + k);
    } else {
        r.hi = *(const uint2*)((const uint16_t*)op + k);
        if constexpr (PREC == LAFF_PREC_FP16X3 || PREC == LAFF_PREC_BF16X3) r.lo = *(const uint2*)((const uint16_t*)op + plane + k);
    }
}
template <int PREC>
__device__ __forceinline__ void operand_from_raw(const OperandRaw& r, float (&x)[4]) {
    if constexpr (PREC == LAFF_PREC_FP32) {
        x[0] = r.f.x; x[1] = r.f.y; x[2] = r.f.z; x[3] = r.f.w;
    } else {
        auto cvt = [](uint16_t b) -> float {
            if constexpr (PREC == LAFF_PREC_BF16 || PREC == LAFF_PREC_BF16X3) return __uint_as_float((unsigned)b << 16);
            else { _Float16 f; __builtin_memcpy(&f, &b, 2); return (float)f; }
        };
        const uint2 o = r.hi;
        x[0] = cvt((uint16_t)o.x); x[1] = cvt((uint16_t)(o.x >> 16)); x[2] = cvt((uint16_t)o.y); x[3] = cvt((uint16_t)(o.y >> 16));
        if constexpr (PREC == LAFF_PREC_FP16X3 || PREC == LAFF_PREC_BF16X3) {
            const uint2 l = r.lo;
            x[0] += cvt((uint16_t)l.x); x[1] += cvt((uint16_t)(l.x >> 16)); x[2] += cvt((uint16_t)l.y); x[3] += cvt((uint16_t)(l.y >> 16));
        }
    }
}

// ONE pass over a row: q^2 = sum_k (x_k - e_k / n_h)^2 with x = operand / prescale and n_h = |e_h| + eps, expanded as
// xx - 2 xe / n + ee / n^2 per head with the three sums in fp64 (the terms cancel to ~1e-8 of their size; fp64 leaves 1e-16), and --
// WITH_GT -- the exact cosine against the ground-truth row v in the same loop.  tt / vv / tv use exactly exact_cos()'s lane map, fma
// order and reduction tree, so the value is bit-identical to what laff_rank_resolve computes for the same two rows.
// EMIT (single-plane 16-bit formats): the operand row is PRODUCED here -- x = fp16 / bf16 (e * prescale), stored to `op` -- instead of read:
// what laff_pack_rows(normalize = 0) would have written for the row (laff_rank_prepare_emit: one launch and one pass over gathered rows
// instead of two).
template <int PREC, bool WITH_GT, bool EMIT = false>
__device__ __forceinline__ float row_pass(const float* __restrict__ e, const void* __restrict__ op, long row, long nrows, int H, int d,
                                          double inv_prescale, int sl, const float* __restrict__ v, double* cos_out, float prescale = 1.0f) {
    const long K = (long)H * d;
    double q2 = 0.0, s = 0.0;
    for (int h = 0; h < H; ++h) {
        const float* eh = e + (long)h * d;
        double tt = 0.0, vv = 0.0, tv = 0.0, xx = 0.0, xe = 0.0;
        // one column group of the lane: a = the embedding's float4, x = the operand's four values, b = the ground-truth row's float4
        auto body = [&](int c, const float4& a, float (&x)[4], const float4& b) {
            if constexpr (EMIT && (PREC == LAFF_PREC_FP16 || PREC == LAFF_PREC_BF16)) {
                const float s4[4] = {a.x * prescale, a.y * prescale, a.z * prescale, a.w * prescale};
                uint16_t b4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (PREC == LAFF_PREC_BF16) { const __bf16 t = (__bf16)s4[i]; x[i] = (float)t; __builtin_memcpy(&b4[i], &t, 2); }
                    else { const _Float16 t = (_Float16)s4[i]; x[i] = (float)t; __builtin_memcpy(&b4[i], &t, 2); }
                }
                *(uint2*)((uint16_t*)const_cast<void*>(op) + row * K + (long)h * d + c) =
                    make_uint2((unsigned)b4[0] | ((unsigned)b4[1] << 16), (unsigned)b4[2] | ((unsigned)b4[3] << 16));
            }
            const double ax = a.x, ay = a.y, az = a.z, aw = a.w;
            tt = fma(ax, ax, tt); tt = fma(ay, ay, tt); tt = fma(az, az, tt); tt = fma(aw, aw, tt);
            if constexpr (WITH_GT) {
                const double bx = b.x, by = b.y, bz = b.z, bw = b.w;
                vv = fma(bx, bx, vv); vv = fma(by, by, vv); vv = fma(bz, bz, vv); vv = fma(bw, bw, vv);
                tv = fma(ax, bx, tv); tv = fma(ay, by, tv); tv = fma(az, bz, tv); tv = fma(aw, bw, tv);
            }
            const double x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
            xx = fma(x0, x0, xx); xx = fma(x1, x1, xx); xx = fma(x2, x2, xx); xx = fma(x3, x3, xx);
            xe = fma(x0, ax, xe); xe = fma(x1, ay, xe); xe = fma(x2, az, xe); xe = fma(x3, aw, xe);
        };
        constexpr bool emits = EMIT && (PREC == LAFF_PREC_FP16 || PREC == LAFF_PREC_BF16);
        constexpr int CH = LAFF_EXACT_CH;
        if (d % (RG * 4 * CH) == 0) {
            // whole batches of CH column groups: every load of a batch is requested before its first value is used (exact_cos.h has the
            // reason: as one rolled / partially unrolled loop hipcc waited for each load, or pair of loads, before issuing the next);
            // the chains run in column order afterwards -- the same fma sequence, the same bits
            for (int c0 = sl * 4; c0 < d; c0 += RG * 4 * CH) {
                float4 a[CH], b[CH];
                OperandRaw o[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int c = c0 + i * (RG * 4);
                    a[i] = *(const float4*)(eh + c);
                    if constexpr (!emits) load_operand_raw<PREC>(op, row * K + (long)h * d + c, nrows * K, o[i]);
                    if constexpr (WITH_GT) b[i] = *(const float4*)(v + (long)h * d + c);
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    float x[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (!emits) operand_from_raw<PREC>(o[i], x);
                    body(c0 + i * (RG * 4), a[i], x, b[i]);
                }
            }
        } else {
#pragma unroll 4
            for (int c = sl * 4; c < d; c += RG * 4) {
                const float4 a = *(const float4*)(eh + c);
                float x[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (!emits) load_operand4<PREC>(op, row * K + (long)h * d + c, nrows * K, x);
                float4 b = make_float4(0, 0, 0, 0);
                if constexpr (WITH_GT) b = *(const float4*)(v + (long)h * d + c);
                body(c, a, x, b);
            }
        }
        tt = group_sum_f64(tt);
        if constexpr (WITH_GT) { vv = group_sum_f64(vv); tv = group_sum_f64(tv); }
        xx = group_sum_f64(xx); xe = group_sum_f64(xe);
        const double n = sqrt(tt) + COS_EPS;
        if constexpr (WITH_GT) s += tv / (n * (sqrt(vv) + COS_EPS));
        const double inv = 1.0 / n;
        q2 += fmax(xx * inv_prescale * inv_prescale - 2.0 * inv * inv_prescale * xe + inv * inv * tt, 0.0);
    }
    if constexpr (WITH_GT) *cos_out = s / (double)H;
    return (float)sqrt(q2);
}

constexpr int PREP_ROWS = 256 / RG;            // text rows per 256-thread block
constexpr int PREP_VROWS = 64;                 // video rows per block: one aligned 64-column group of the GEMM (one band value per wave)

template <int PREC>
__global__ __launch_bounds__(256) void rank_prepare_kernel(const float* __restrict__ Et, const float* __restrict__ Ev,
                                                           const void* __restrict__ T, const void* __restrict__ V, int Nt, int Nv,
                                                           int H, int d, float inv_prescale, float unit, float c_acc, const int* __restrict__ gt_col,
                                                           int col0, double* __restrict__ s_gt64, float* __restrict__ band_t,
                                                           float* __restrict__ band_v, int* __restrict__ zero_count,
                                                           unsigned* __restrict__ pairs, long vblocks, int emit, float prescale) {
    __shared__ float blkmax[PREP_ROWS];
    const int sl = threadIdx.x & (RG - 1), grp = threadIdx.x / RG;
    const long K = (long)H * d;
    // vblocks = ceil(Nv / PREP_VROWS) video blocks come first in the grid (they are the longer ones); 0 when only the text side runs
    if (blockIdx.x == 0 && threadIdx.x < 4 && pairs) pairs[threadIdx.x] = 0u;       // pair counter + overflow flag
    const float rsqrt_h = 1.0f / sqrtf((float)H);
    if ((long)blockIdx.x >= vblocks) {
        const long t0 = ((long)blockIdx.x - vblocks) * PREP_ROWS + grp;
        const bool ok = t0 < Nt;
        const long t = ok ? t0 : Nt - 1;                       // idle groups shadow the last row (shuffles stay convergent)
        if (ok && zero_count && sl == 0) zero_count[t] = 0;
        const float* e = Et + t * K;
        const int c = gt_col[t] - col0;
        const bool own = c >= 0 && c < Nv;
        double sg = -INFINITY;
        float q;
        if (emit & 1) {                                             // kernel-uniform: the text operand is produced here (every group writes
            if (__builtin_amdgcn_ballot_w64(own) != 0ull) {        // its own row; idle groups shadow row Nt - 1 with the same values)
                q = row_pass<PREC, true, true>(e, T, t, Nt, H, d, (double)inv_prescale, sl, Ev + (long)(own ? c : 0) * K, &sg, prescale);
                if (!own) sg = -INFINITY;
            } else {
                q = row_pass<PREC, false, true>(e, T, t, Nt, H, d, (double)inv_prescale, sl, nullptr, nullptr, prescale);
            }
        } else if (__builtin_amdgcn_ballot_w64(own) != 0ull) {     // wave-uniform: some group of this wavefront owns its column
            q = row_pass<PREC, true>(e, T, t, Nt, H, d, (double)inv_prescale, sl, Ev + (long)(own ? c : 0) * K, &sg);
            if (!own) sg = -INFINITY;
        } else {
            q = row_pass<PREC, false>(e, T, t, Nt, H, d, (double)inv_prescale, sl, nullptr, nullptr);
        }
        if (ok && sl == 0) {
            s_gt64[t] = sg;
            band_t[t] = q * rsqrt_h * (1.0f + unit) * 1.0001f + c_acc;
        }
    } else {
        // video rows [64 b, 64 b + 64): per-row band_v and, behind the Nv per-row values, the maximum of the block (what a wave of the
        // GEMM epilogue uses for its 64 columns)
        const long vb = (long)blockIdx.x;
        float mx = 0.0f;
        for (int k = 0; k < PREP_VROWS / PREP_ROWS; ++k) {
            const long v0 = vb * PREP_VROWS + k * PREP_ROWS + grp;
            const bool ok = v0 < Nv;
            const long v = ok ? v0 : Nv - 1;
            const float q = (emit & 2) ? row_pass<PREC, false, true>(Ev + v * K, V, v, Nv, H, d, (double)inv_prescale, sl, nullptr, nullptr, prescale)
                                       : row_pass<PREC, false>(Ev + v * K, V, v, Nv, H, d, (double)inv_prescale, sl, nullptr, nullptr);
            const float b = q * rsqrt_h * 1.0001f;
            if (ok && sl == 0) band_v[v] = b;
            if (ok) mx = fmaxf(mx, b);
        }
        if (sl == 0) blkmax[grp] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = 0.0f;
            for (int i = 0; i < PREP_ROWS; ++i) m = fmaxf(m, blkmax[i]);
            band_v[((Nv + 3) & ~3) + vb] = m;          // block maxima start at a 16-byte aligned offset (the GEMM fetches them by DMA)
        }
    }
}

hipError_t launch_rank_prepare(const float* Et, const float* Ev, const void* T, const void* V, int Nt, int Nv, int H, int d,
                               int precision, float prescale, const int* gt_col, int col0, double* s_gt64, float* band_t,
                               float* band_v, int* zero_count, unsigned* pairs, int sides, hipStream_t st, int emit) {
    // sides: 1 = the text rows (s_gt64, band_t, cleared count / list header), 2 = the video rows (band_v), 3 = both
    const long vblocks = (sides & 2) ? ((long)Nv + PREP_VROWS - 1) / PREP_VROWS : 0;
    const long grid = ((sides & 1) ? ((long)Nt + PREP_ROWS - 1) / PREP_ROWS : 0) + vblocks;
    if (grid == 0) return hipSuccess;
    if (grid <= 0 || grid > 0x7fffffffL) return hipErrorInvalidValue;
    const float inv = 1.0f / prescale;
    // fp32 accumulation of the exact products: K terms (3K for a hi/lo split, plus its dropped lo*lo term <= 2^-22), 2^-23 each
    // (covers round-to-nearest and truncating accumulators), + 2^-20 for the fp32 copy of s_gt64, the scaling and the rounding of the
    // accumulator-unit thresholds the GEMM epilogue compares against
    const bool x3 = precision == LAFF_PREC_FP16X3 || precision == LAFF_PREC_BF16X3;
    const float c_acc = (float)((double)H * d * (x3 ? 3.0 : 1.0) * 1.1920929e-7 + 9.5367432e-7 + (x3 ? 2.3841858e-7 : 0.0));
#define LAFF_PREP(P, U)                                                                                                          \
    hipLaunchKernelGGL((rank_prepare_kernel<P>), dim3((unsigned)grid), dim3(256), 0, st, Et, Ev, T, V, Nt, Nv, H, d, inv, U, c_acc, gt_col, \
                       col0, s_gt64, band_t, band_v, zero_count, pairs, vblocks, emit, prescale)
    switch (precision) {
        case LAFF_PREC_FP32: LAFF_PREP(LAFF_PREC_FP32, 5.9604645e-8f); break;
        case LAFF_PREC_FP16: LAFF_PREP(LAFF_PREC_FP16, 4.8828125e-4f); break;
        case LAFF_PREC_BF16: LAFF_PREP(LAFF_PREC_BF16, 3.90625e-3f); break;
        case LAFF_PREC_FP16X3: LAFF_PREP(LAFF_PREC_FP16X3, 4.8828125e-4f); break;
        case LAFF_PREC_BF16X3: LAFF_PREP(LAFF_PREC_BF16X3, 3.90625e-3f); break;
        default: return hipErrorInvalidValue;
    }
#undef LAFF_PREP
    return hipGetLastError();
}

constexpr unsigned RESOLVE_QCAP = 512;           // queued pairs per wavefront (LDS)

// ---- evaluation.eval (/root/reference/evaluation.py:92-109) by ONE workgroup, as the tail of the resolve launch -------------------
// The block that draws the last ticket of laff_rank_resolve_metrics sees every count final (the other blocks released theirs before
// taking a ticket) and turns them into ranks and the seven metrics itself: one launch and one ~6 us launch gap less than
// laff_rank_resolve + laff_rank_metrics, no scratch round trips between blocks, and the 64 result bytes go straight into the caller's
// pinned buffer when the device can address it (no copy node).  Same definitions as rank_metrics_kernel below; the fp64 sum of
// reciprocals is taken in this block's own fixed order (thread-strided partials, xor-shuffle tree, waves in order), so it agrees with
// the multi-block kernel to rounding (1e-16 relative), not bit for bit.
struct MetricsTail {
    int n;                 // 0: no tail (plain laff_rank_resolve)
    int base;
    int* ranks_out;        // [n] or null
    double* out8;          // device: 7 metrics + flag
    double* host8;         // the same 8 doubles in device-addressable host memory, or null
    unsigned* ticket;      // RESOLVE_TICKET_BYTES, zero before the launch; left zero
};
constexpr unsigned RESOLVE_TICKET_GROUPS = 32;
constexpr size_t RESOLVE_TICKET_BYTES = 256 * (1 + RESOLVE_TICKET_GROUPS);
struct MetricsLds {
    double shd[16];
    unsigned long long shl[16][4];
    int shm[16][2];
    int wsum[4];
    int sel[2];
    unsigned hist[512 * 9];
};
constexpr size_t RESOLVE_POOL_BYTES = sizeof(MetricsLds) > 4 * RESOLVE_QCAP * 8 ? sizeof(MetricsLds) : 4 * RESOLVE_QCAP * 8;

template <int NT>
__device__ void metrics_single_block(const int* r, int n, int base, int* ranks_out, double* out8, double* host8, MetricsLds& L,
                                     bool force_bad = false) {
    static_assert(NT == 256, "select_bin below wants exactly four wavefronts");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // One CU does all of this with one wavefront per SIMD, so the per-rank work is kept to a handful of integer instructions (the
    // first version -- reciprocal, sums and two histogram updates per rank -- took 19 us for 40,000 ranks).  L.hist is cut in three:
    //   lo  [256 x 9]  ranks 1..255, eight replicas per bin (pitch 9): nearly all ranks of a retrieval that works; v == 1 (41 % at C4;
    //                  64 lanes adding to one LDS word serialise) is counted in a register instead;
    //   mid [2048]     ranks 256..2303, one word per rank;
    //   hi  [256]      ranks from 2304 up by their top byte (bin min(v >> 8, 255)), and invalid ranks (< 1): the long path -- fp64
    //                  reciprocal, sum, min / max per rank.
    // R@1/5/10, the mean rank and the mean reciprocal rank of the ranks below 2304 are read off the 2,303 bins afterwards.
    unsigned* const lo = L.hist;
    unsigned* const mid = L.hist + 256 * 9;
    unsigned* const hi = L.hist + 256 * 9 + 2048;
    constexpr unsigned MID0 = 256u, MID1 = 2304u;
    for (int i = tid; i < 512 * 9; i += NT) L.hist[i] = 0;
    __syncthreads();
    unsigned n1 = 0, nbig = 0;
    unsigned long long sum = 0;
    double isum = 0;
    int mx = 0, mn = 0x7fffffff;
    const int rep8 = lane & 7;
    auto take = [&](int v) {
        const unsigned u = (unsigned)v;
        if (u - 1u < MID1 - 1u) {                                 // 1 <= v < 2304
            if (u == 1u) ++n1;
            else atomicAdd(u < MID0 ? &lo[u * 9 + rep8] : &mid[u - MID0], 1u);
        } else {
            ++nbig;
            sum += (unsigned long long)(long long)v;
            const double d = (double)v;
            double q = (double)__builtin_amdgcn_rcpf((float)v);      // 1 / v: fp32 reciprocal + two Newton steps in fp64
            q = q * (2.0 - d * q);
            q = q * (2.0 - d * q);
            isum += q;
            mx = max(mx, v); mn = min(mn, v);
            atomicAdd(&hi[min(u >> 8, 255u)], 1u);
        }
    };
    // The counts were last touched by the other workgroups' device-scope atomics (performed at the memory side: the XCDs' L2s are not
    // coherent with each other) and nobody has read them with plain loads during this launch, so after the caller's acquire fence (an
    // L2 invalidate of this XCD, no write-back) plain 16-byte loads fetch them from memory.
    const bool vec = ((((uintptr_t)r) | ((uintptr_t)ranks_out)) & 15) == 0;
    const int n4 = vec ? n >> 2 : 0;
    // batches of two 16-byte loads per lane (the launch must stay within 80 VGPRs), the NEXT batch requested before the current one is consumed
    auto each = [&](auto&& f) {
        constexpr int B = 2;
        const int nb = n4 / (B * NT);                             // full batches
        int4 cur[B], nxt[B];
        if (nb > 0) {
#pragma unroll
            for (int u = 0; u < B; ++u) cur[u] = ((const int4*)r)[tid + u * NT];
        }
        for (int b = 0; b < nb; ++b) {
            const int i4 = b * B * NT + tid;
            if (b + 1 < nb) {
#pragma unroll
                for (int u = 0; u < B; ++u) nxt[u] = ((const int4*)r)[i4 + (B + u) * NT];
            }
#pragma unroll
            for (int u = 0; u < B; ++u) {
                const int i = 4 * (i4 + u * NT);
                f(i, cur[u].x + base); f(i + 1, cur[u].y + base); f(i + 2, cur[u].z + base); f(i + 3, cur[u].w + base);
            }
#pragma unroll
            for (int u = 0; u < B; ++u) cur[u] = nxt[u];
        }
        for (int i4 = nb * B * NT + tid; i4 < n4; i4 += NT) {
            const int4 q = ((const int4*)r)[i4];
            f(4 * i4, q.x + base); f(4 * i4 + 1, q.y + base); f(4 * i4 + 2, q.z + base); f(4 * i4 + 3, q.w + base);
        }
        for (int i = 4 * n4 + tid; i < n; i += NT) f(i, r[i] + base);
    };
    each([&](int i, int v) {
        if (ranks_out) ranks_out[i] = v;           // (the four stores of a 16-byte group are merged by the compiler: consecutive i)
        take(v);
    });
    __syncthreads();                                              // (every wave's histogram updates are in)
    // lo: the replicas of bin tid -> one word; the sums over the bins: thread tid takes lo bin tid and mid bins 8 tid .. 8 tid + 7
    unsigned own_lo = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) own_lo += lo[tid * 9 + q];
    unsigned msum = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned c = mid[8 * tid + q], v = MID0 + 8u * (unsigned)tid + (unsigned)q;
        msum += c;
        sum += (unsigned long long)c * v;
        if (c) isum += (double)c / (double)v;
    }
    // (32 consecutive threads hold one 256-rank block of mid[]: its total is what hi[1 + tid / 32] lacks)
    unsigned blk = msum;
    blk = wave_allreduce<16>(blk, [](unsigned a, unsigned b) { return a + b; });      // (32 consecutive lanes each)
    unsigned c5 = 0, c10 = 0;
    if (tid == 1) own_lo = 0;                                     // (lo[1] is n1, added below)
    if (tid >= 2) {
        c5 = tid <= 5 ? own_lo : 0u; c10 = tid <= 10 ? own_lo : 0u;
        sum += (unsigned long long)own_lo * (unsigned)tid;
        if (own_lo) isum += (double)own_lo / (double)tid;
    }
    unsigned nmid = msum;
    n1 = wave_allsum(n1); nbig = wave_allsum(nbig); nmid = wave_allsum(nmid); sum = wave_allsum(sum);
    c5 = wave_allsum(c5); c10 = wave_allsum(c10);
    isum = wave_allsum(isum);
    mx = wave_allmax(mx); mn = wave_allmin(mn);
    if (lane == 0) {
        L.shl[wave][0] = n1 | ((unsigned long long)nbig << 32); L.shl[wave][1] = c5 | ((unsigned long long)c10 << 32);
        L.shl[wave][2] = nmid; L.shl[wave][3] = sum;
        L.shd[wave] = isum;
        L.shm[wave][0] = mx; L.shm[wave][1] = mn;
    }
    __syncthreads();                                              // (all replicas read: lo[] is re-used as the compact histograms)
    // compact layout from here on: L.hist[0..255] = lo, L.hist[256..511] = hi (mid[] stays where it is, beyond 256 * 9)
    L.hist[tid] = own_lo;
    {
        const unsigned h = hi[tid];
        __syncthreads();                                          // (hi[] read before L.hist[256 + tid] is written: the regions do not overlap, but keep the order explicit)
        L.hist[256 + tid] = h;
    }
    __syncthreads();
    if ((tid & 31) == 0) L.hist[256 + 1 + tid / 32] += blk;       // hi[1..8] += the mid blocks
    n1 = nbig = nmid = c5 = c10 = 0; sum = 0; isum = 0; mx = 0; mn = 0x7fffffff;
    for (int w = 0; w < NT / 64; ++w) {
        n1 += (unsigned)L.shl[w][0]; nbig += (unsigned)(L.shl[w][0] >> 32); c5 += (unsigned)L.shl[w][1]; c10 += (unsigned)(L.shl[w][1] >> 32);
        nmid += (unsigned)L.shl[w][2]; sum += L.shl[w][3]; isum += L.shd[w];
        mx = max(mx, L.shm[w][0]); mn = min(mn, L.shm[w][1]);
    }
    sum += n1;
    isum += (double)n1;
    c5 += n1; c10 += n1;
    __syncthreads();
    if (tid == 0) { L.hist[1] = n1; L.hist[256] = (unsigned)n - nbig - nmid; }      // lo[1] and hi[0] = #{v < 256} were counted in registers
    __syncthreads();
    if ((n > 0 && mn < 1) || force_bad) {                         // invalid input (or a list that overflowed): flag + NaN metrics
        if (tid < 7) { out8[tid] = __builtin_nan(""); if (host8) host8[tid] = __builtin_nan(""); }
        if (tid == 7) { out8[7] = 1.0; if (host8) host8[7] = 1.0; }
        return;
    }
    if (nbig == 0) mx = (int)MID1 - 1;                            // (only the general select below reads mx, and only for ranks >= 65,280)
    const int k = n / 2;                                          // k-th smallest, 0-based
    auto select_bin = [&](int own, int less) {
        int inc = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) L.wsum[wave] = inc;
        __syncthreads();
        int before = less + inc - own;
        for (int w = 0; w < wave; ++w) before += L.wsum[w];
        if (before <= k && k < before + own) { L.sel[0] = tid; L.sel[1] = before; }     // exactly one thread
        __syncthreads();
    };
    unsigned prefix = 0, mask = 0;
    int less = 0, shift, below = 0;
    bool below_known = false;
    const int in_lo = (int)L.hist[256];                           // hi[0] = #{v < 256}
    if (k < in_lo) {
        select_bin((int)L.hist[tid], 0);
        prefix = (unsigned)L.sel[0]; less = L.sel[1];
        mask = 0xffffffffu;
        shift = -8;
        for (int b = (int)prefix - 1; b >= 1; --b)
            if (L.hist[b]) { below = b; break; }
        below_known = true;
    } else {
        select_bin((int)L.hist[256 + tid], 0);
        const int top = L.sel[0];
        if (top >= 1 && top <= 8) {
            // the median's 256-rank block lies inside mid[]: its low byte comes out of those bins, no pass over the ranks
            less = L.sel[1];
            select_bin((int)mid[256 * (top - 1) + tid], less);
            prefix = ((unsigned)top << 8) | (unsigned)L.sel[0]; less = L.sel[1];
            mask = 0xffffffffu;
            shift = -8;
        } else if (top < 255) { prefix = (unsigned)top << 8; less = L.sel[1]; mask = 0xffffff00u; shift = 0; }
        else shift = ((32 - __clz(mx | 1) + 7) / 8 - 1) * 8;
    }
    for (; shift >= 0; shift -= 8) {
        __syncthreads();
        for (int i = tid; i < 256 * 9; i += NT) L.hist[i] = 0;
        __syncthreads();
        each([&](int, int vi) {
            const unsigned v = (unsigned)vi;
            if ((v & mask) == prefix) atomicAdd(&L.hist[((v >> shift) & 255u) * 9 + rep8], 1u);
        });
        __syncthreads();
        int own = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) own += (int)L.hist[tid * 9 + q];
        select_bin(own, less);
        prefix |= (unsigned)L.sel[0] << shift;
        mask |= 255u << shift;
        less = L.sel[1];
    }
    const int med_lo = (int)prefix;
    double med = med_lo;
    if (n > 0 && (n & 1) == 0 && less >= k) {
        if (!below_known) {
            each([&](int, int v) {
                if (v < med_lo) below = max(below, v);
            });
            below = wave_allmax(below);
            __syncthreads();
            if (lane == 0) L.shm[wave][0] = below;
            __syncthreads();
            below = 0;
            for (int w = 0; w < NT / 64; ++w) below = max(below, L.shm[w][0]);
        }
        med = 0.5 * (med + (double)below);
    }
    if (tid == 0) {
        const double dn = (double)n;
        double o[8] = {100.0 * ((double)n1 / dn), 100.0 * ((double)c5 / dn), 100.0 * ((double)c10 / dn), floor(med), (double)sum / dn,
                       isum / dn, isum / dn, 0.0};
#pragma unroll
        for (int i = 0; i < 8; ++i) { out8[i] = o[i]; if (host8) host8[i] = o[i]; }       // (visible to the host when the launch completes)
    }
}

// A list that overflowed: flag in the header + count[0] pushed below every legitimate value, both as DEVICE-SCOPE atomics -- a plain
// store may sit dirty in this XCD's L2 where neither the fused metrics tail's finishing workgroup (another XCD, no release fence on
// this side) nor the next launch's atomics on count[0] would meet it.  (The add survives concurrent +1 updates of count[0].)
__device__ __forceinline__ void poison_overflow(unsigned* pairs, int* count) {
    __hip_atomic_store(pairs + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(count, -(1 << 26), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the overflow condition of either list format, from the header the GEMM launch left (complete before this launch started)
__device__ __forceinline__ bool list_overflowed(const unsigned* pairs, unsigned pair_cap) {
    if (pairs[2] & 0x80000000u) {
        const unsigned taken = pairs[0], NW = pairs[2] & 0x7fffffffu, NCH = pairs[3];
        return pairs[1] != 0u || NW > NCH || taken > NCH - (NW < NCH ? NW : NCH);
    }
    const unsigned n_over = pairs[0], regA = pairs[2];
    const unsigned long long room = pair_cap > regA ? pair_cap - regA : 0u;
    return pairs[1] != 0u || n_over > room;
}

// ---- the strip kernel's list (sim_strip.hip): header {chunks taken from the pool, overflow flag, NW | 1 << 31, NCH} | NCH per-chunk
// entry counts (rounded up to 4 words) | NCH chunks of STRIP_CHUNK entries of STRIP_ENTRY_WORDS words.  Chunks 0 .. NW - 1 belong to
// the GEMM's wavefronts, chunks NW .. NW + taken - 1 were taken from the pool; the first count[c] entries of chunk c are valid.
// An entry = {row, colbase, lo, hi | mask16, gt, 0, 0 | x[16]}: the 16 raw accumulators a lane of the GEMM held for `row` (columns
// colbase + 8 (e >> 2) + (e & 3)), the accumulator-unit thresholds the GEMM counted against (x > hi was counted there), the elements
// that may be listed (columns beyond the matrix excluded) and the row's ground-truth column (shard-local): that element is never
// listed, and S takes the exact score there.  A 16-lane group takes an entry, lane e tests element e (lo <= x <= hi: exactly the band
// test of the tiled kernel's epilogue); the pairs inside the band are queued and re-scored like the pairs of the other list format.
__device__ __forceinline__ void resolve_groups(const float* __restrict__ Et, const float* __restrict__ Ev, int H, int d,
                                               const double* __restrict__ s_gt64, int* __restrict__ count, float* __restrict__ S,
                                               long lds, unsigned* __restrict__ pairs, unsigned pair_cap,
                                               unsigned (*queue)[RESOLVE_QCAP][2]) {
    const int sl = threadIdx.x & (RG - 1);
    const long K = (long)H * d;
    const unsigned taken = pairs[0], NW = pairs[2] & 0x7fffffffu, NCH = pairs[3];
    const unsigned cnt_words = (NCH + 3u) & ~3u;
    if (pairs[1] != 0u || NW > NCH || taken > NCH - (NW < NCH ? NW : NCH)) {            // the pool ran out: entries were dropped
        if (blockIdx.x == 0 && threadIdx.x == 0) poison_overflow(pairs, count);
    }
    const unsigned long long nchunks = (unsigned long long)NW + taken < NCH ? (unsigned long long)NW + taken : NCH;
    const unsigned long long slots = nchunks * STRIP_CHUNK;
    const unsigned* entries = pairs + 4 + cnt_words;
    (void)pair_cap;
    auto one = [&](unsigned r, unsigned c, bool ok) {
#ifdef LAFF_RESOLVE_HOTROWS
        const double ex = exact_cos(Et + (long)(r & 63u) * K, Ev + (long)(c & 63u) * K, H, d, sl);
#else
        const double ex = exact_cos(Et + (long)r * K, Ev + (long)c * K, H, d, sl);
#endif
        const double sg = s_gt64[r];
        const bool above = ex > sg;
        if (ok && sl == 0) {
            if (above) atomicAdd(count + r, 1);
            if (S) {
                float f = (float)ex;
                const float sgf = (float)sg;
                if (above && !(f > sgf)) f = nextafterf(sgf, INFINITY);
                S[(long)r * lds + c] = f;
            }
        }
    };
    constexpr unsigned QCAP = RESOLVE_QCAP;
    const unsigned wv = threadIdx.x >> 6, sub = (threadIdx.x / RG) & 3u;
    const unsigned group = (blockIdx.x * 256u + threadIdx.x) / RG, ngroups = gridDim.x * (256u / RG);
    unsigned qn = 0;                                                         // wave-uniform
    auto drain = [&]() {
#ifndef LAFF_RESOLVE_SCAN_ONLY
        for (unsigned i = 0; i < qn; i += 4) {
            const unsigned j = i + sub;
            const bool ok = j < qn;
            const unsigned k = ok ? j : qn - 1;
            one(queue[wv][k][0], queue[wv][k][1], ok);
        }
#endif
        qn = 0;
    };
    const unsigned long long trips = (slots + ngroups - 1) / ngroups;
    const unsigned gpb = 256u / RG;                                          // groups per block
    for (unsigned long long it = 0; it < trips; ++it) {                      // wave-uniform trip count
        if (qn > QCAP - 64) drain();
        // every block walks a CONTIGUOUS range of entries: the dumps of one GEMM wavefront share their 64 text rows, which then come
        // out of this XCD's L2 the second time (entries dealt round-robin over the whole launch: 59.7 us at C4, this way 56.6)
        const unsigned long long idx = ((unsigned long long)blockIdx.x * trips + it) * gpb + (group % gpb);
        const bool live = idx < slots && (unsigned)(idx % STRIP_CHUNK) < pairs[4 + (unsigned)(idx / STRIP_CHUNK)];
        bool inb = false;
        unsigned row = 0, col = 0;
        if (live) {
            const unsigned* e = entries + idx * STRIP_ENTRY_WORDS;
            const uint4 h0 = *(const uint4*)e;                               // the 16 lanes of the group read the same 16 bytes
            const unsigned mask16 = e[4], gt_col = e[5];
            const float x = __uint_as_float(e[8 + sl]);
            const float lo = __uint_as_float(h0.z), hi = __uint_as_float(h0.w);
            row = h0.x;
            col = h0.y + 8u * ((unsigned)sl >> 2) + ((unsigned)sl & 3u);
            if (col == gt_col) {
                // the ground-truth entry of this row: never listed; S takes the exact score there (the GEMM stored its own value)
                if (S) S[(long)row * lds + col] = (float)s_gt64[row];
            } else {
                inb = ((mask16 >> sl) & 1u) && __builtin_amdgcn_fmed3f(x, lo, hi) == x;
            }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(inb);
        if (inb) {
            const unsigned at = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            queue[wv][at][0] = row;
            queue[wv][at][1] = col;
        }
        qn += (unsigned)__builtin_popcountll(m);
    }
    drain();
}

// one 16-lane group per listed pair.  The list (written by the banded GEMM epilogue): header {n_overflow, overflow flag, A, chunk},
// then A slots in per-wavefront segments of `chunk` slots (valid pairs first, the rest marked row = 0xffffffff), then n_overflow
// pairs appended with the counter.  A 16-lane group walks a segment until the first invalid slot; the overflow region is shared out
// four pairs per wavefront at a time.  count[row] += 1 when the exact score beats the exact ground-truth score; S (optional) takes the fp32 value of the
// exact score, nudged by one ulp where rounding to fp32 would hide a strict inequality, so that ranks recounted from S
// (laff_rank_count) equal the ranks produced here.  More pairs than the list holds: overflow flag + count[0] poisoned with -(2^26) (rank < 1 trips
// the error flag of laff_rank_metrics*).
__device__ __forceinline__ void resolve_pairs(const float* __restrict__ Et, const float* __restrict__ Ev, int H, int d,
                                              const double* __restrict__ s_gt64, int* __restrict__ count, float* __restrict__ S,
                                              long lds, unsigned* __restrict__ pairs, unsigned pair_cap,
                                              unsigned (*queue)[RESOLVE_QCAP][2]);

// (six wavefronts per SIMD = the launch's 6 x CUs workgroups in one resident round: at most 80 VGPRs)
#ifndef LAFF_RESOLVE_WAVES
#define LAFF_RESOLVE_WAVES 5      // 94 registers with 8 row loads in flight per 16-lane group (6 = 80 registers: spills)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LAFF_RESOLVE_WAVES))) void rank_resolve_kernel(
    const float* __restrict__ Et, const float* __restrict__ Ev, int H, int d, const double* __restrict__ s_gt64, int* __restrict__ count,
    float* __restrict__ S, long lds, unsigned* __restrict__ pairs, unsigned pair_cap, MetricsTail mt) {
    __shared__ __attribute__((aligned(16))) unsigned char pool[RESOLVE_POOL_BYTES];      // the pair queues, then the metrics tail's state
    unsigned (*queue)[RESOLVE_QCAP][2] = reinterpret_cast<unsigned (*)[RESOLVE_QCAP][2]>(pool);
    // read before block 0 raises pairs[1]: every workgroup derives the same answer from what the GEMM launch left in the header, so the
    // finishing workgroup of the metrics tail does not depend on seeing block 0's poison
    const bool overflowed = list_overflowed(pairs, pair_cap);
    if (pairs[2] & 0x80000000u)             // the strip kernel's list (sim_strip.hip): dumped groups of 16 raw accumulators
        resolve_groups(Et, Ev, H, d, s_gt64, count, S, lds, pairs, pair_cap, queue);
    else
        resolve_pairs(Et, Ev, H, d, s_gt64, count, S, lds, pairs, pair_cap, queue);
    if (mt.n <= 0) return;
    // ---- metrics tail: every block releases its counts and draws a ticket; the last one sees all of them
    // (No fences: a release / acquire pair at agent scope is an L2 write-back + invalidate per workgroup -- with 1,536 of them the launch
    // took 0.34 ms instead of 0.06.  What the last block needs from the others are their count updates, device-scope atomics that have
    // been performed once s_waitcnt vmcnt(0) returns; the ticket is a relaxed device-scope atomic issued after that wait, and the last
    // block reads the counts behind the control dependency on its ticket and ONE acquire fence of its own.)
    // Tickets in two levels (same-address device atomics serialise at ~10 ns each: 1,536 of them on one word, arriving together at the
    // end of a balanced launch, cost 14 us): workgroup b draws from counter 1 + b % 32 (each on its own 256-byte line); the last
    // arrival of a group draws from counter 0.
    __shared__ unsigned s_ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x % RESOLVE_TICKET_GROUPS, gsz = (gridDim.x - g + RESOLVE_TICKET_GROUPS - 1u) / RESOLVE_TICKET_GROUPS;
        const unsigned ngroups = gridDim.x < RESOLVE_TICKET_GROUPS ? gridDim.x : RESOLVE_TICKET_GROUPS;
        unsigned* const tg = mt.ticket + 64u * (1u + g);
        unsigned last = 0u;
        LAFF_TAIL_RELEASE();
        if (__hip_atomic_fetch_add(tg, 1u, LAFF_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == gsz - 1u) {
            __hip_atomic_store(tg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(mt.ticket, 1u, LAFF_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1u) {
                __hip_atomic_store(mt.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1u;
            }
        }
        s_ticket = last;
    }
    __syncthreads();
    if (!s_ticket) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // one L2 invalidate in one workgroup (no write-back)
    metrics_single_block<256>(count, mt.n, mt.base, mt.ranks_out, mt.out8, mt.host8, *reinterpret_cast<MetricsLds*>(pool), overflowed);
}

// the tiled kernel's list: header {n_overflow, overflow flag, A, chunk}, A slots in per-wavefront segments, then the overflow pairs
__device__ __forceinline__ void resolve_pairs(const float* __restrict__ Et, const float* __restrict__ Ev, int H, int d,
                                              const double* __restrict__ s_gt64, int* __restrict__ count, float* __restrict__ S,
                                              long lds, unsigned* __restrict__ pairs, unsigned pair_cap,
                                              unsigned (*queue)[RESOLVE_QCAP][2]) {
    const int sl = threadIdx.x & (RG - 1);
    const long K = (long)H * d;
    constexpr unsigned QCAP = RESOLVE_QCAP;
    const unsigned n_over = pairs[0], regA = pairs[2];
    const unsigned long long room = pair_cap > regA ? pair_cap - regA : 0u;
    if (n_over > room) {
        // the poison survives an int32 all-reduce(SUM) over up to 16 shards (laff_amd/dist.py 'video' scheme): 16 * -(2^26) = -2^30 does
        // not wrap, and no legitimate count (< 2^26 videos) lifts it back above zero
        if (blockIdx.x == 0 && threadIdx.x == 0) poison_overflow(pairs, count);
    }
    auto one = [&](unsigned r, unsigned c, bool ok) {
        const double ex = exact_cos(Et + (long)r * K, Ev + (long)c * K, H, d, sl);
        const double sg = s_gt64[r];
        const bool above = ex > sg;
        if (ok && sl == 0) {
            if (above) atomicAdd(count + r, 1);
            if (S) {
                float f = (float)ex;
                const float sgf = (float)sg;
                if (above && !(f > sgf)) f = nextafterf(sgf, INFINITY);
                S[(long)r * lds + c] = f;
            }
        }
    };
    // Two steps per wavefront, so that the expensive part always runs four pairs wide:
    //   scan : its four 16-lane groups read blocks of 4 consecutive slots (segments are multiples of 4 slots with their valid pairs
    //          first; the overflow region follows the segments) and queue the valid pairs in LDS (ballot + mbcnt, wave-private);
    //   drain: the queue is re-scored four pairs at a time, one per group.
    const unsigned wv = threadIdx.x >> 6, sub = (threadIdx.x / RG) & 3u, lane = threadIdx.x & 63u;
    const unsigned n = (unsigned)(n_over < room ? n_over : room);
    const unsigned long long total = (unsigned long long)regA + n;
    const unsigned group = (blockIdx.x * 256u + threadIdx.x) / RG, ngroups = gridDim.x * (256u / RG);
    unsigned qn = 0;                                                         // wave-uniform
    auto drain = [&]() {
        for (unsigned i = 0; i < qn; i += 4) {
            const unsigned j = i + sub;
            const bool ok = j < qn;
            const unsigned k = ok ? j : qn - 1;
            one(queue[wv][k][0], queue[wv][k][1], ok);
        }
        qn = 0;
    };
    const unsigned long long trips = (total + 4ull * ngroups - 1) / (4ull * ngroups);
    for (unsigned long long it = 0; it < trips; ++it) {                      // wave-uniform trip count
        if (qn > QCAP - 16) drain();
        const unsigned long long s0 = 4ull * (it * ngroups + group);
        uint4 p0 = make_uint4(0xffffffffu, 0, 0xffffffffu, 0), p1 = p0;
        if (s0 < total) {
            p0 = *(const uint4*)(pairs + 4 + 2 * s0);                        // {r0, c0, r1, c1}
            if (p0.x != 0xffffffffu && s0 + 2 < total) p1 = *(const uint4*)(pairs + 4 + 2 * s0 + 4);   // {r2, c2, r3, c3}
        }
        const unsigned rs[4] = {p0.x, p0.z, p1.x, p1.z}, cs[4] = {p0.y, p0.w, p1.y, p1.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool v = sl == 0 && s0 + k < total && rs[k] != 0xffffffffu;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
            if (v) {
                const unsigned at = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                queue[wv][at][0] = rs[k];
                queue[wv][at][1] = cs[k];
            }
            qn += (unsigned)__builtin_popcountll(m);
        }
    }
    (void)lane;
    drain();
}

// ---- the listed pairs of a VIDEO shard's banded GEMM, exported for the owners of the TEXT rows (laff_amd/dist.py, 'video16': the
// fp32 text rows never leave their owner, so the exact re-score of a pair happens there).  Reads either list format exactly like the
// resolve kernel (for the strip kernel's dumps: the band test, the ground-truth entry of S <- the exact score), and appends every pair
// as {row - bounds[o], col + col0} to the bucket of the owner o of its text row (bounds[o] <= row < bounds[o + 1]): `out` holds `world`
// buckets of `cap` slots, pre-filled with 0xffffffff -- each bucket, behind a 4-word header {0, 0, cap, 4}, is a list laff_rank_resolve
// reads (valid pairs first).  One atomic per (wavefront drain, owner).  A full bucket: flag + count[0] poisoned like an overflowing list.
struct ExportArgs {
    const int* bounds;     // [world + 1] first text row of every owner
    int world;
    int col0;
    unsigned* out;         // [world][cap][2]
    unsigned cap;
    unsigned* fill;        // [world + 1]: pairs appended per bucket, [world] = overflow flag
};
constexpr unsigned EXPORT_QCAP = 512;
__global__ __launch_bounds__(256) void rank_export_kernel(const double* __restrict__ s_gt64, int* __restrict__ count, float* __restrict__ S,
                                                          long lds, unsigned* __restrict__ pairs, unsigned pair_cap, ExportArgs x) {
    __shared__ unsigned queue[4][EXPORT_QCAP][3];
    __shared__ int sbounds[17];
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const int sl = threadIdx.x & (RG - 1);
    if (threadIdx.x <= (unsigned)x.world) sbounds[threadIdx.x] = x.bounds[threadIdx.x];
    __syncthreads();
    auto owner = [&](unsigned row) {
        int o = 0;
        for (int k = 1; k < x.world; ++k) o += (int)row >= sbounds[k] ? 1 : 0;
        return (unsigned)o;
    };
    unsigned qn = 0;                                                         // wave-uniform
    auto poison = [&]() { __hip_atomic_store(x.fill + x.world, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); poison_overflow(pairs, count); };
    auto drain = [&]() {
        for (int o = 0; o < x.world; ++o) {
            unsigned n_o = 0;
            for (unsigned i = 0; i < qn; i += 64) {
                const bool hit = i + lane < qn && queue[wv][i + lane][2] == (unsigned)o;
                n_o += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit));
            }
            if (n_o == 0) continue;
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(x.fill + o, n_o);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + n_o > x.cap && lane == 0) poison();
            unsigned run = 0;
            for (unsigned i = 0; i < qn; i += 64) {
                const bool hit = i + lane < qn && queue[wv][i + lane][2] == (unsigned)o;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (hit) {
                    const unsigned slot = base + run + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (slot < x.cap) {
                        unsigned* dst = x.out + ((size_t)o * x.cap + slot) * 2;
                        dst[0] = queue[wv][i + lane][0] - (unsigned)sbounds[o];
                        dst[1] = queue[wv][i + lane][1] + (unsigned)x.col0;
                    }
                }
                run += (unsigned)__builtin_popcountll(m);
            }
        }
        qn = 0;
    };
    auto push = [&](bool v, unsigned row, unsigned col) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
        if (v) {
            const unsigned at = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            queue[wv][at][0] = row;
            queue[wv][at][1] = col;
            queue[wv][at][2] = owner(row);
        }
        qn += (unsigned)__builtin_popcountll(m);
    };
    const unsigned group = (blockIdx.x * 256u + threadIdx.x) / RG, ngroups = gridDim.x * (256u / RG);
    if (pairs[2] & 0x80000000u) {
        // the strip kernel's list: dumped groups of 16 raw accumulators (see resolve_groups)
        const unsigned taken = pairs[0], NW = pairs[2] & 0x7fffffffu, NCH = pairs[3];
        const unsigned cnt_words = (NCH + 3u) & ~3u;
        if ((pairs[1] != 0u || NW > NCH || taken > NCH - (NW < NCH ? NW : NCH)) && blockIdx.x == 0 && threadIdx.x == 0) poison();
        const unsigned long long nchunks = (unsigned long long)NW + taken < NCH ? (unsigned long long)NW + taken : NCH;
        const unsigned long long slots = nchunks * STRIP_CHUNK;
        const unsigned* entries = pairs + 4 + cnt_words;
        const unsigned long long trips = (slots + ngroups - 1) / ngroups;
        for (unsigned long long it = 0; it < trips; ++it) {
            if (qn > EXPORT_QCAP - 64) drain();
            const unsigned long long idx = it * ngroups + group;
            const bool live = idx < slots && (unsigned)(idx % STRIP_CHUNK) < pairs[4 + (unsigned)(idx / STRIP_CHUNK)];
            bool inb = false;
            unsigned row = 0, col = 0;
            if (live) {
                const unsigned* e = entries + idx * STRIP_ENTRY_WORDS;
                const uint4 h0 = *(const uint4*)e;
                const unsigned mask16 = e[4], gt_col = e[5];
                const float v = __uint_as_float(e[8 + sl]);
                const float lo = __uint_as_float(h0.z), hi = __uint_as_float(h0.w);
                row = h0.x;
                col = h0.y + 8u * ((unsigned)sl >> 2) + ((unsigned)sl & 3u);
                if (col == gt_col) {
                    if (S) S[(long)row * lds + col] = (float)s_gt64[row];
                } else {
                    inb = ((mask16 >> sl) & 1u) && __builtin_amdgcn_fmed3f(v, lo, hi) == v;
                }
            }
            push(inb, row, col);
        }
    } else {
        const unsigned n_over = pairs[0], regA = pairs[2];
        const unsigned long long room = pair_cap > regA ? pair_cap - regA : 0u;
        if (n_over > room && blockIdx.x == 0 && threadIdx.x == 0) poison();
        const unsigned n = (unsigned)(n_over < room ? n_over : room);
        const unsigned long long total = (unsigned long long)regA + n;
        const unsigned long long nthreads = (unsigned long long)gridDim.x * 256u;
        const unsigned long long trips = (total + nthreads - 1) / nthreads;
        for (unsigned long long it = 0; it < trips; ++it) {
            if (qn > EXPORT_QCAP - 64) drain();
            const unsigned long long i = it * nthreads + (unsigned long long)blockIdx.x * 256u + threadIdx.x;
            unsigned row = 0xffffffffu, col = 0;
            if (i < total) { row = pairs[4 + 2 * i]; col = pairs[4 + 2 * i + 1]; }
            push(row != 0xffffffffu, row, col);
        }
    }
    drain();
}

// (a kernel, not hipMemsetAsync: inside a captured HIP graph the two memset nodes of this call did not re-run reliably on replay --
// the fill counters kept growing: tools/debug/dbg_v16_rccl.py)
__global__ __launch_bounds__(256) void rank_export_init_kernel(uint4* __restrict__ out, size_t n16, unsigned* __restrict__ fill, int world) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (size_t i = i0; i < n16; i += step) out[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    if (i0 <= (size_t)world) fill[i0] = 0u;
}

hipError_t launch_rank_export(const double* s_gt64, int* count, float* S, int lds, unsigned* pairs, unsigned pair_cap, const int* bounds,
                              int world, int col0, unsigned* out, unsigned cap, unsigned* fill, hipStream_t st) {
    if (world < 1 || world > 16 || (cap & 3)) return hipErrorInvalidValue;
    const size_t n16 = (size_t)world * cap / 2;                   // 16-byte units: cap % 4 == 0, out is 16-byte aligned (api.hip)
    hipLaunchKernelGGL(rank_export_init_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, (size_t)(8 * g_num_cus))), dim3(256), 0, st,
                       (uint4*)out, n16, fill, world);
    ExportArgs x{bounds, world, col0, out, cap, fill};
    hipLaunchKernelGGL(rank_export_kernel, dim3((unsigned)(4 * g_num_cus)), dim3(256), 0, st, s_gt64, count, S, (long)lds, pairs, pair_cap, x);
    return hipGetLastError();
}

hipError_t launch_rank_resolve(const float* Et, const float* Ev, int Nt, int Nv, int H, int d, const double* s_gt64, int* count,
                               float* S, int lds, unsigned* pairs, unsigned pair_cap, hipStream_t st, int metrics_n, int base,
                               int* ranks_out, double* out8, double* host8, unsigned* ticket) {
    (void)Nt; (void)Nv;
    // one resident round: ~19 KiB of LDS and ~70 VGPRs per block admit 7 blocks per CU; 6 x CUs leaves a margin (a second, sparse
    // round of the 2,048-block grid doubled this launch's time)
    MetricsTail mt{metrics_n, base, ranks_out, out8, host8, ticket};
    hipLaunchKernelGGL(rank_resolve_kernel, dim3((unsigned)(LAFF_RESOLVE_WAVES * g_num_cus)), dim3(256), 0, st, Et, Ev, H, d, s_gt64, count, S, (long)lds, pairs,
                       pair_cap, mt);
    return hipGetLastError();
}

// evaluation.eval (/root/reference/evaluation.py:92-109) for single-GT rows, on the device.
// out7 = r1, r5, r10, medr, meanr, mir, mAP (= mir).  err[0] = 1.0 (and NaN metrics) if a rank < 1 was seen, else 0.0.
// rank = r[i] + base (base = 1 turns the fused "better-scoring videos" counts into ranks); ranks_out, when given, receives them.
//
// Up to MS_GMAX blocks of 1024 threads, one rank per thread per trip (one workgroup is one CU: 40,000 reciprocals and histogram
// updates on a single CU took 23-31 us).  Every block leaves its partial sums in its own slot of the scratch (reduced by the last
// block in a fixed order: the fp64 reciprocal sum does not depend on arrival order) and adds two 256-bin histograms to the scratch
// with integer atomics: lo[v] for v < 256, hi[v >> 8] (hi[255]: everything from 65,280 up).  The block that draws the last ticket
// finishes: the median is an order statistic of positive integers -- retrieval ranks are small, so it usually falls into lo[] and
// nothing is read again; otherwise hi[] fixes the top byte and ONE pass of that block over the ranks (MSB-first radix select with
// 8-bit digits, histogram bins replicated 32 times in LDS: ranks pile up on a few values and 64 lanes adding to one LDS word
// serialise) the low byte; ranks from 65,280 up take the general select from the top non-zero byte.  The last block clears the
// scratch for the next launch.
constexpr int MS_GMAX = 256;
struct MetricsPartial { unsigned long long c1, c5, c10, sum; double isum; int mx, mn; int pad[4]; };      // 64 bytes
static_assert(sizeof(MetricsPartial) == 64, "scratch layout");
// scratch words: {ticket, 3 x pad} | lo[256] | hi[256] | pad to 4 KiB | MS_GMAX partials
constexpr size_t MS_BYTES = 4096 + (size_t)MS_GMAX * sizeof(MetricsPartial);
// (+ the ticket lines of laff_rank_resolve_metrics behind it: the partials above are not left zero)
size_t rank_metrics_scratch_bytes() { return MS_BYTES + ((RESOLVE_TICKET_BYTES + 4095) & ~(size_t)4095); }
size_t rank_resolve_ticket_offset() { return MS_BYTES; }

__global__ __launch_bounds__(1024) void rank_metrics_kernel(const int* __restrict__ r, int n, int base, int* __restrict__ ranks_out,
                                                            double* __restrict__ out7, double* __restrict__ err,
                                                            unsigned* __restrict__ scratch, double* __restrict__ host8) {
    __shared__ double shd[16];
    __shared__ unsigned long long shl[16][4];
    __shared__ int shm[16][2];
    __shared__ unsigned hist[256 * 33];
    __shared__ int wsum[4];
    __shared__ int sel[2];
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = (int)gridDim.x;
    unsigned* const ghist = scratch + 4;
    MetricsPartial* const parts = (MetricsPartial*)((char*)scratch + 4096);

    // ---- every block: its share of the sums and of the two histograms (bins replicated 8 times in LDS, pitch 9)
    for (int i = tid; i < 512 * 9; i += 1024) hist[i] = 0;
    __syncthreads();
    unsigned long long c1 = 0, c5 = 0, c10 = 0, sum = 0;
    double isum = 0;
    int mx = 0, mn = 0x7fffffff;
    const int rep8 = lane & 7;
    for (long i = (long)blockIdx.x * 1024 + tid; i < n; i += (long)G * 1024) {
        const int v = r[i] + base;
        if (ranks_out) ranks_out[i] = v;
        c1 += v <= 1; c5 += v <= 5; c10 += v <= 10;
        sum += (unsigned long long)(long long)v;
        // 1/v: fp32 reciprocal refined by two Newton steps in fp64 (relative error < 1e-16) instead of a full fp64 division
        const double d = (double)v;
        double q = (double)__builtin_amdgcn_rcpf((float)v);
        q = q * (2.0 - d * q);
        q = q * (2.0 - d * q);
        isum += q;
        mx = max(mx, v); mn = min(mn, v);
        const unsigned u = (unsigned)v;
        if (u < 256u) atomicAdd(&hist[u * 9 + rep8], 1u);
        atomicAdd(&hist[(256u + min(u >> 8, 255u)) * 9 + rep8], 1u);
    }
    auto block_sums = [&]() {                                     // -> thread 0 holds the block's totals
        c1 = wave_allsum(c1); c5 = wave_allsum(c5); c10 = wave_allsum(c10); sum = wave_allsum(sum);
        isum = wave_allsum(isum);
        mx = wave_allmax(mx); mn = wave_allmin(mn);
        if (lane == 0) {
            shl[wave][0] = c1; shl[wave][1] = c5; shl[wave][2] = c10; shl[wave][3] = sum;
            shd[wave] = isum;
            shm[wave][0] = mx; shm[wave][1] = mn;
        }
        __syncthreads();
        if (tid == 0) {
            c1 = c5 = c10 = sum = 0; isum = 0; mx = 0; mn = 0x7fffffff;
            for (int w = 0; w < 16; ++w) {
                c1 += shl[w][0]; c5 += shl[w][1]; c10 += shl[w][2]; sum += shl[w][3]; isum += shd[w];
                mx = max(mx, shm[w][0]); mn = min(mn, shm[w][1]);
            }
        }
    };
    block_sums();
    // What the last block needs from the others travels as device-scope (memory-side) stores and atomics, has been performed once
    // s_waitcnt vmcnt(0) returns, and is read back with device-scope loads: no fence anywhere (an agent-scope release / acquire is a
    // write-back / invalidate of the XCD's whole L2 -- three of them per block were most of this launch's 20 us).
    if (tid == 0) {
        unsigned long long* q = (unsigned long long*)(parts + blockIdx.x);
        const unsigned long long w[6] = {c1, c5, c10, sum, (unsigned long long)__double_as_longlong(isum),
                                         (unsigned long long)(unsigned)mx | ((unsigned long long)(unsigned)mn << 32)};
#pragma unroll
        for (int k = 0; k < 6; ++k) __hip_atomic_store(q + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 512) {
        unsigned t = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += hist[tid * 9 + q];
        if (t) __hip_atomic_fetch_add(ghist + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        LAFF_TAIL_RELEASE();
        s_ticket = __hip_atomic_fetch_add(scratch, 1u, LAFF_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_ticket != (unsigned)(G - 1)) return;
    LAFF_TAIL_ACQUIRE();

    // ---- the last block
    {
        c1 = c5 = c10 = sum = 0; isum = 0; mx = 0; mn = 0x7fffffff;
        if (tid < G) {
            const unsigned long long* q = (const unsigned long long*)(parts + tid);
            unsigned long long w[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) w[k] = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c1 = w[0]; c5 = w[1]; c10 = w[2]; sum = w[3];
            isum = __longlong_as_double((long long)w[4]);
            mx = (int)(unsigned)w[5]; mn = (int)(unsigned)(w[5] >> 32);
        }
        __syncthreads();                                          // (shl / shd / shm are reused)
        block_sums();
        if (tid == 0) { shl[0][0] = c1; shl[0][1] = c5; shl[0][2] = c10; shl[0][3] = sum; shd[0] = isum; shm[0][0] = mx; shm[0][1] = mn; }
    }
    // the two histograms -> LDS (hist[0..255] = lo, hist[256..511] = hi), scratch cleared for the next launch
    if (tid < 512) {
        hist[tid] = __hip_atomic_load(ghist + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ghist + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) __hip_atomic_store(scratch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    mx = shm[0][0]; mn = shm[0][1];
    if (n > 0 && mn < 1) {                                        // invalid input: report (flag + NaN metrics), do not select
        if (tid < 7) { out7[tid] = __builtin_nan(""); if (host8) host8[tid] = __builtin_nan(""); }
        if (tid == 0) { err[0] = 1.0; if (host8) host8[7] = 1.0; }
        return;
    }
    const int k = n / 2;                                          // k-th smallest, 0-based
    // the ranks again, one workgroup: 16-byte loads, eight of them in flight (every pass is latency-bound)
    const bool vec = (((uintptr_t)r) & 15) == 0;
    const int n4 = vec ? n >> 2 : 0;
    auto each = [&](auto&& f) {
#pragma unroll 8
        for (int i4 = tid; i4 < n4; i4 += 1024) {
            const int4 q = ((const int4*)r)[i4];
            f(q.x + base); f(q.y + base); f(q.z + base); f(q.w + base);
        }
        for (int i = 4 * n4 + tid; i < n; i += 1024) f(r[i] + base);
    };
    // owner of position k among 256 bins held by threads 0..255 (`own` each): sel[0] = bin, sel[1] = #{elements in lower bins} + less
    auto select_bin = [&](int own, int less) {
        int inc = own;
        if (tid < 256) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            if (lane == 63) wsum[wave] = inc;
        }
        __syncthreads();
        if (tid < 256) {
            int before = less + inc - own;
            for (int w = 0; w < wave; ++w) before += wsum[w];
            if (before <= k && k < before + own) { sel[0] = tid; sel[1] = before; }     // exactly one thread
        }
        __syncthreads();
    };
    unsigned prefix = 0, mask = 0;
    int less = 0;                                                 // #{r < smallest value with the current prefix}
    int shift;
    int below = 0;                                                // largest rank below the median's value, when known from lo[]
    bool below_known = false;
    const int in_lo = (int)hist[256];                             // hi[0] = #{v < 256}
    if (k < in_lo) {
        select_bin(tid < 256 ? (int)hist[tid] : 0, 0);
        prefix = (unsigned)sel[0]; less = sel[1];
        mask = 0xffffffffu;
        shift = -8;
        for (int b = (int)prefix - 1; b >= 1; --b)                // (uniform: every thread walks the same LDS words)
            if (hist[b]) { below = b; break; }
        below_known = true;
    } else {
        select_bin(tid < 256 ? (int)hist[256 + tid] : 0, 0);
        if (sel[0] < 255) { prefix = (unsigned)sel[0] << 8; less = sel[1]; mask = 0xffffff00u; shift = 0; }
        else shift = ((32 - __clz(mx | 1) + 7) / 8 - 1) * 8;      // ranks from 65,280 up: the general select
    }
    const int rep = lane & 31;
    for (; shift >= 0; shift -= 8) {
        __syncthreads();
        for (int i = tid; i < 256 * 33; i += 1024) hist[i] = 0;
        __syncthreads();
        each([&](int vi) {
            const unsigned v = (unsigned)vi;
            if ((v & mask) == prefix) atomicAdd(&hist[((v >> shift) & 255u) * 33 + rep], 1u);
        });
        __syncthreads();
        int own = 0;
        if (tid < 256) {
#pragma unroll
            for (int q = 0; q < 32; ++q) own += (int)hist[tid * 33 + q];
        }
        select_bin(own, less);
        prefix |= (unsigned)sel[0] << shift;
        mask |= 255u << shift;
        less = sel[1];
    }
    const int med_lo = (int)prefix;
    double med = med_lo;
    if (n > 0 && (n & 1) == 0 && less >= k) {
        // sorted[k-1] equals sorted[k] unless exactly k elements are smaller: then it is the largest of those
        if (!below_known) {
            each([&](int v) {
                if (v < med_lo) below = max(below, v);
            });
            below = wave_allmax(below);
            __syncthreads();
            if (lane == 0) shm[wave][0] = below;
            __syncthreads();
            below = 0;
            for (int w = 0; w < 16; ++w) below = max(below, shm[w][0]);
        }
        med = 0.5 * (med + (double)below);
    }
    if (tid == 0) {
        const double dn = (double)n;
        const double o[7] = {100.0 * ((double)shl[0][0] / dn), 100.0 * ((double)shl[0][1] / dn), 100.0 * ((double)shl[0][2] / dn), floor(med),
                             (double)shl[0][3] / dn, shd[0] / dn, shd[0] / dn};
#pragma unroll
        for (int i = 0; i < 7; ++i) { out7[i] = o[i]; if (host8) host8[i] = o[i]; }
        err[0] = 0.0;
        if (host8) host8[7] = 0.0;                                // (device-addressable host memory: visible when the launch completes)
    }
}

hipError_t launch_rank_metrics(const int* r, int n, int base, int* ranks_out, double* out7, double* err, unsigned* scratch, hipStream_t st,
                               double* host8) {
    const int G = std::min(MS_GMAX, std::max(1, (n + 1023) / 1024));
    hipLaunchKernelGGL(rank_metrics_kernel, dim3((unsigned)G), dim3(1024), 0, st, r, n, base, ranks_out, out7, err, scratch, host8);
    return hipGetLastError();
}

hipError_t launch_gather_gt(const float* S, int Nt, int Nv, int lds, const int* gt_col, int col0, float* s_gt,
                            hipStream_t st) {
    hipLaunchKernelGGL(gather_gt_kernel, dim3((Nt + 255) / 256), dim3(256), 0, st, S, Nt, Nv, (long)lds, gt_col, col0, s_gt);
    return hipGetLastError();
}

hipError_t launch_rank_count(const float* S, int Nt, int Nv, int lds, const int* gt_col, int col0, const float* s_gt,
                             int* count, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(rank_count_kernel, dim3(Nt), dim3(256), 0, st, S, Nt, Nv, (long)lds, gt_col, col0, s_gt, count,
                       accumulate);
    return hipGetLastError();
}

// ---- exact video-to-text positions (predictor.py:262-270 ranks every column of the score matrix) ---------------------------------
// count[t] = #{ t' != t : exact(t', v) > exact(t, v) },  v = the owner column of t, on a score matrix S whose entries are within
// band_t[t'] + band_v[v] of exact(t', v) (what laff_sim_gemm_banded / laff_rank_resolve leave behind; the bound is laff_rank_prepare's).
// A column is compared with the exact scores of ITS captions (s_gt64, one threshold per caption, G per pass): an entry further than
// its band from a threshold is decided on S; the others (a few per caption) go to a list {t', v, t} and are re-scored in fp64 from
// the fp32 embeddings by v2t_resolve_kernel with the arithmetic of exact_cos(), i.e. of the text-to-video ranks.
// Grid (column blocks of 32, row chunks): every block adds its partial counts (count is cleared by the caller).
template <int G>
__global__ __launch_bounds__(256) void v2t_band_count_kernel(const float* __restrict__ S, int Nt, int Nv, long lds,
                                                             const int* __restrict__ grp_off, const int* __restrict__ grp_idx,
                                                             const double* __restrict__ s_gt64, const float* __restrict__ band_t,
                                                             const float* __restrict__ band_v, int* __restrict__ count,
                                                             unsigned* __restrict__ list, unsigned cap, int pass, int rows_per_block) {
    constexpr int CT = 32;
    __shared__ float thr[CT][G + 1];
    __shared__ int own[CT][G + 1];
    __shared__ int cnts[8][CT][G + 1];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * CT + cx;
    const bool col_ok = c < Nv;
    int g0 = 0, gn = 0;
    if (col_ok) {
        g0 = grp_off[c] + pass * G;
        gn = min(max(grp_off[c + 1] - g0, 0), G);
    }
    if (ry == 0) {
        for (int i = 0; i < G; ++i) {
            const int t = (i < gn) ? grp_idx[g0 + i] : -1;
            own[cx][i] = t;
            thr[cx][i] = (t >= 0) ? (float)s_gt64[t] : INFINITY;          // (the rounding is inside the band: rank.hip, c_acc)
        }
    }
    __syncthreads();
    float th[G];
    int ow[G], cn[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        th[i] = thr[cx][i];
        ow[i] = own[cx][i];
        cn[i] = 0;
    }
    if (col_ok && gn > 0) {
        const float bv = band_v[c];
        const int r_end = min(Nt, ((int)blockIdx.y + 1) * rows_per_block);
        for (int r = (int)blockIdx.y * rows_per_block + ry; r < r_end; r += 8) {
            const float v = S[(long)r * lds + c];
            const float e = (band_t[r] + bv) * 1.000001f + 1e-9f;          // (fp32 roundings of the sum and of v - th)
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const float dlt = v - th[i];
                cn[i] += (dlt > e);
                if (fabsf(dlt) <= e && r != ow[i] && ow[i] >= 0) {
                    const unsigned slot = atomicAdd(list, 1u);
                    if (slot < cap) {
                        list[4 + 3 * (size_t)slot] = (unsigned)r;
                        list[5 + 3 * (size_t)slot] = (unsigned)c;
                        list[6 + 3 * (size_t)slot] = (unsigned)ow[i];
                    } else {
                        list[1] = 1u;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < G; ++i) cnts[ry][cx][i] = cn[i];
    __syncthreads();
    if (ry == 0 && col_ok) {
        for (int i = 0; i < gn; ++i) {
            int tot = 0;
#pragma unroll
            for (int y = 0; y < 8; ++y) tot += cnts[y][cx][i];
            if (tot) atomicAdd(count + grp_idx[g0 + i], tot);
        }
    }
}

__global__ __launch_bounds__(256) void v2t_resolve_kernel(const float* __restrict__ Et, const float* __restrict__ Ev, int H, int d,
                                                          const double* __restrict__ s_gt64, int* __restrict__ count,
                                                          const unsigned* __restrict__ list, unsigned cap) {
    const int sl = threadIdx.x & (RG - 1);
    const long K = (long)H * d;
    const unsigned n = min(list[0], cap);
    const unsigned group = (blockIdx.x * 256u + threadIdx.x) / RG, ngroups = gridDim.x * (256u / RG);
    const unsigned trips = (n + ngroups - 1) / ngroups;
    for (unsigned it = 0; it < trips; ++it) {                                 // wave-uniform trip count (the shuffles of exact_cos)
        const unsigned j = it * ngroups + group;
        const bool ok = j < n;
        const unsigned k = ok ? j : n - 1;
        const unsigned r = list[4 + 3 * (size_t)k], c = list[5 + 3 * (size_t)k], t = list[6 + 3 * (size_t)k];
        const double ex = exact_cos(Et + (long)r * K, Ev + (long)c * K, H, d, sl);
        if (ok && sl == 0 && ex > s_gt64[t]) atomicAdd(count + t, 1);
    }
}

hipError_t launch_v2t_count_exact(const float* S, int Nt, int Nv, int lds, const int* grp_off, const int* grp_idx, int max_group,
                                  const float* Et, const float* Ev, int H, int d, const double* s_gt64, const float* band_t,
                                  const float* band_v, int* count, unsigned* list, unsigned cap, hipStream_t st) {
    hipError_t e = hipMemsetAsync(count, 0, (size_t)Nt * sizeof(int), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(list, 0, 16, st);
    if (e != hipSuccess) return e;
    const unsigned gx = (unsigned)((Nv + 31) / 32);
    // row chunks: enough blocks for four rounds of the chip, at least 256 rows each
    int gy = (int)std::min<long>(std::max<long>(1, (8L * g_num_cus + gx - 1) / gx), std::max<long>(1, Nt / 256));
    const int rows_per_block = ((Nt + gy - 1) / gy + 7) & ~7;
    gy = (Nt + rows_per_block - 1) / rows_per_block;
    const int G = max_group <= 4 ? 4 : (max_group <= 8 ? 8 : 16);
    const int passes = (max_group + G - 1) / G;
    for (int p = 0; p < passes; ++p) {
#define LAFF_V2T(GG)                                                                                                             \
    hipLaunchKernelGGL((v2t_band_count_kernel<GG>), dim3(gx, (unsigned)gy), dim3(256), 0, st, S, Nt, Nv, (long)lds, grp_off, grp_idx,    \
                       s_gt64, band_t, band_v, count, list, cap, p, rows_per_block)
        switch (G) {
            case 4: LAFF_V2T(4); break;
            case 8: LAFF_V2T(8); break;
            default: LAFF_V2T(16); break;
        }
#undef LAFF_V2T
    }
    hipLaunchKernelGGL(v2t_resolve_kernel, dim3((unsigned)(4 * g_num_cus)), dim3(256), 0, st, Et, Ev, H, d, s_gt64, count, list, cap);
    return hipGetLastError();
}

hipError_t launch_v2t_count(const float* S, int Nt, int Nv, int lds, const int* grp_off, const int* grp_idx,
                            int max_group, int* count, hipStream_t st) {
    const unsigned grid = (unsigned)((Nv + 31) / 32);
    int G = max_group <= 4 ? 4 : (max_group <= 8 ? 8 : (max_group <= 16 ? 16 : 32));
    const int passes = (max_group + G - 1) / G;
    for (int p = 0; p < passes; ++p) {
        switch (G) {
            case 4: hipLaunchKernelGGL((v2t_count_kernel<4>), dim3(grid), dim3(256), 0, st, S, Nt, Nv, (long)lds, grp_off, grp_idx, count, p); break;
            case 8: hipLaunchKernelGGL((v2t_count_kernel<8>), dim3(grid), dim3(256), 0, st, S, Nt, Nv, (long)lds, grp_off, grp_idx, count, p); break;
            case 16: hipLaunchKernelGGL((v2t_count_kernel<16>), dim3(grid), dim3(256), 0, st, S, Nt, Nv, (long)lds, grp_off, grp_idx, count, p); break;
            default: hipLaunchKernelGGL((v2t_count_kernel<32>), dim3(grid), dim3(256), 0, st, S, Nt, Nv, (long)lds, grp_off, grp_idx, count, p); break;
        }
    }
    return hipGetLastError();
}

}  // namespace laff

// ---- top-K per text row --------------------------------------------------------------------
// The result writers of the reference (/root/reference/predictor.py:53-88) take `np.argsort(S, axis=1)[i][::-1][0:TopK]`
// of every row (TopK = 500 / 2000).  Here: one 256-thread workgroup per row; the row lives in LDS as order-preserving
// 32-bit keys; the K-th largest is found by an MSB-first radix select (4 passes of a 256-bin LDS histogram); the K
// survivors are compacted and bitonic-sorted as 64-bit (key, index) pairs.  Order: score descending, ties by index
// descending (what a stable ascending argsort read backwards yields).
namespace laff {

__device__ __forceinline__ unsigned f2key(float f) {
    unsigned b = __float_as_uint(f);
    if (b == 0x80000000u) b = 0u;                    // -0.0 == +0.0 for argsort: one key, ties resolved by index
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

// among elements i with pred(i), find the byte-wise prefix of the `need`-th largest value of val(i);
// returns the full 32-bit value; *above = how many selected elements are strictly greater.
template <typename Val, typename Pred>
__device__ unsigned radix_select(int n, int need, Val val, Pred pred, unsigned* hist, int* sh, int* above_out) {
    unsigned prefix = 0, mask = 0;
    int above = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (pred(i)) {
                const unsigned v = val(i);
                if ((v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255], 1u);
            }
        __syncthreads();
        if (threadIdx.x == 0) {              // 256 bins: a serial scan from the top is cheap enough
            int acc = above, b = 255;
            for (; b > 0; --b) {
                if (acc + (int)hist[b] >= need) break;
                acc += hist[b];
            }
            sh[0] = b;
            sh[1] = acc;
        }
        __syncthreads();
        prefix |= (unsigned)sh[0] << shift;
        mask |= 255u << shift;
        above = sh[1];
        __syncthreads();
    }
    *above_out = above;
    return prefix;
}

template <int KP /* power of two >= K */>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ S, int Nv, long lds, int K,
                                                        int* __restrict__ idx_out, float* __restrict__ val_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned* keys = (unsigned*)smem_raw;                               // [Nv]
    unsigned long long* sel = (unsigned long long*)(keys + ((Nv + 3) & ~3));   // [KP]
    unsigned* hist = (unsigned*)(sel + KP);                             // [256]
    int* sh = (int*)(hist + 256);                                       // [4]
    const long t = blockIdx.x;
    const float* row = S + t * lds;
    for (int i = threadIdx.x; i < Nv; i += 256) keys[i] = f2key(row[i]);
    for (int i = threadIdx.x; i < KP; i += 256) sel[i] = 0ull;          // padding sorts last
    __syncthreads();
    int above = 0;
    const unsigned T = radix_select(Nv, K, [&](int i) { return keys[i]; }, [](int) { return true; }, hist, sh, &above);
    // ties at the threshold: take the (K - above) LARGEST indices among keys == T
    const int need_eq = K - above;
    int above_i = 0;
    const unsigned Ti = radix_select(Nv, need_eq, [](int i) { return (unsigned)i; }, [&](int i) { return keys[i] == T; }, hist, sh, &above_i);
    if (threadIdx.x == 0) sh[2] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < Nv; i += 256) {
        const unsigned k = keys[i];
        if (k > T || (k == T && (unsigned)i >= Ti)) {
            const int p = atomicAdd(&sh[2], 1);
            if (p < KP) sel[p] = ((unsigned long long)k << 32) | (unsigned)i;
        }
    }
    __syncthreads();
    // bitonic sort, descending, KP elements
    for (int k2 = 2; k2 <= KP; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < KP; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = sel[i], b = sel[l];
                    const bool desc = (i & k2) == 0;
                    if (desc ? a < b : a > b) { sel[i] = b; sel[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < K; i += 256) {
        const unsigned long long e = sel[i];
        idx_out[t * K + i] = (int)(unsigned)e;
        val_out[t * K + i] = key2f((unsigned)(e >> 32));
    }
}

hipError_t launch_topk_rows(const float* S, int Nt, int Nv, int lds, int K, int* idx_out, float* val_out, hipStream_t st) {
    if (K > 8192) return hipErrorInvalidValue;
#define LAFF_TOPK(P)                                                                                                     \
    do {                                                                                                                 \
        const size_t smem = (size_t)((Nv + 3) & ~3) * 4 + (size_t)(P) * 8 + 256 * 4 + 16;                                  \
        if (smem > 160 * 1024) return hipErrorInvalidValue;                                                              \
        if (smem > 64 * 1024) {                                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)topk_rows_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            if (e != hipSuccess) return e;                                                                               \
        }                                                                                                                \
        hipLaunchKernelGGL((topk_rows_kernel<P>), dim3(Nt), dim3(256), smem, st, S, Nv, (long)lds, K, idx_out, val_out);  \
    } while (0)
    if (K <= 64) LAFF_TOPK(64);
    else if (K <= 512) LAFF_TOPK(512);
    else if (K <= 2048) LAFF_TOPK(2048);
    else if (K <= 4096) LAFF_TOPK(4096);
    else LAFF_TOPK(8192);
#undef LAFF_TOPK
    return hipGetLastError();
}

}  // namespace laff
