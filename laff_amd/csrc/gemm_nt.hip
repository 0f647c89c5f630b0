// gemm_nt.hip -- the one dense-contraction kernel of the LAFF hot path on gfx950 (MI355X, CDNA4).
//
//   out[r][c] = epilogue( sum_k R[r][k] * C[c][k] )            ("NT": both operands K-contiguous)
//
// used as
//   * FC projection  (TransformNet.forward, /root/reference/model/model.py:257-276):  R = X[N,Dk], C = W[D,Dk],
//     fp32 MFMA (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain), epilogue bias -> act -> folded BN;
//   * similarity     (loss.cosine_sim + get_txt2vis_matrix, loss.py:30-34, model/model.py:1003-1016):
//     R = T[Nt,K], C = V[Nv,K], fp16/bf16 MFMA (v_mfma_f32_32x32x16_*), 1 or 3 passes (hi/lo split),
//     epilogue scale (+ fused ground-truth rank count, predictor.py:232-244 in count form).
//
// Structure (v1): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA
// tiles of 32x32), K-step = 128 bytes per row (64 halves / 32 floats), two LDS stages of 32 KiB
// (2 workgroups per CU), operands staged global -> LDS with 16-byte direct-to-LDS loads
// (global_load_lds_dwordx4).  The LDS image is lane-linear, so the bank-conflict swizzle is applied to the
// per-lane SOURCE address and undone on the ds_read_b128 side (chunk ^= (row>>1)&7 inside each 128-byte row).
// The MFMA "A" operand is fed from C rows and "B" from R rows so that each lane ends up with 4 consecutive
// output columns of one output row per accumulator quad -> 16-byte global stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace laff {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0, 0, 0, 0};

constexpr int TILE = 128;          // output tile edge
constexpr int ROWB = 128;          // bytes of K per row per K-step
constexpr int OPB = TILE * ROWB;   // 16 KiB per operand per stage
constexpr int STAGEB = 2 * OPB;    // 32 KiB per stage

template <int MODE> struct ModeTraits;
template <> struct ModeTraits<GEMM_F32> { static constexpr int ESZ = 4; };
template <> struct ModeTraits<GEMM_F16> { static constexpr int ESZ = 2; };
template <> struct ModeTraits<GEMM_BF16> { static constexpr int ESZ = 2; };

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case 1: return tanhf(v);
        case 2: return fmaxf(v, 0.0f);
        case 3: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// ---- staging ----------------------------------------------------------------------------------------------
// One operand tile = 128 rows x 8 chunks of 16 B.  LDS slot p (16-B units) = row*8 + cs holds source chunk
// c = cs ^ ((row>>1)&7) of that row.
template <bool GLDS>
__device__ __forceinline__ void stage_operand(const char* __restrict__ base, int row0, int nrows, long ldb /*bytes*/,
                                              long kbyte0, long kbytes_valid /*bytes of K in this segment*/,
                                              char* lds_op, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int p = it * 256 + tid;            // 0..1023
        const int row = p >> 3;
        const int cs = p & 7;
        const int c = cs ^ ((row >> 1) & 7);
        int gr = row0 + row;
        gr = gr < nrows ? gr : nrows - 1;        // clamp: garbage rows are never stored
        const long kb = kbyte0 + (long)c * 16;
        if constexpr (GLDS) {
            const char* src = (kb + 16 <= kbytes_valid) ? base + (long)gr * ldb + kb : (const char*)g_zero16;
            // wave-uniform LDS base + lane*16: slot index of lane 0 of this wave for this round
            char* dst = lds_op + (size_t)(it * 256 + (tid & ~63)) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else {
            // generic path: element-wise bounds (K tail / unaligned rows), through registers
            uint32_t v[4] = {0, 0, 0, 0};
            const char* rowp = base + (long)gr * ldb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long b = kb + 4 * j;
                if (b + 4 <= kbytes_valid) {
                    v[j] = *(const uint32_t*)(rowp + b);
                } else if (b < kbytes_valid) {          // 16-bit tail: one valid half
                    v[j] = *(const uint16_t*)(rowp + b);
                }
            }
            *(uint4*)(lds_op + (size_t)p * 16) = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
}

__device__ __forceinline__ uint4 lds_frag(const char* lds_op, int row, int chunk) {
    const int cs = chunk ^ ((row >> 1) & 7);
    return *(const uint4*)(lds_op + row * ROWB + cs * 16);
}

// ---- kernel -----------------------------------------------------------------------------------------------
template <int MODE, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ESZ = ModeTraits<MODE>::ESZ;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;

    // ---- block -> tile: XCD-contiguous chunks (bijective), then groups of 8 tile rows swept along c
    const int tiles_r = (a.nR + TILE - 1) / TILE, tiles_c = (a.nC + TILE - 1) / TILE;
    const int nb = tiles_r * tiles_c;
    int lin;
    {
        const int bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int gsz_full = 8 * tiles_c;
    const int grp = lin / gsz_full;
    const int first_r = grp * 8;
    const int gsz = min(tiles_r - first_r, 8);
    const int in_grp = lin - grp * gsz_full;
    const int tile_r = first_r + in_grp % gsz;
    const int tile_c = in_grp / gsz;
    const int r0 = tile_r * TILE, c0 = tile_c * TILE;

    const long ldRb = (long)a.ldR * ESZ, ldCb = (long)a.ldC * ESZ;
    const long Kb = (long)a.K * ESZ;
    const int kt_per_seg = (int)((Kb + ROWB - 1) / ROWB);
    const int nkt = kt_per_seg * a.nseg;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    auto stage = [&](int kt, int buf) {
        const int seg = kt / kt_per_seg;
        const long kb0 = (long)(kt - seg * kt_per_seg) * ROWB;
        char* s = smem + buf * STAGEB;
        stage_operand<GLDS>((const char*)a.R + a.segR[seg], r0, a.nR, ldRb, kb0, Kb, s, tid);
        stage_operand<GLDS>((const char*)a.C + a.segC[seg], c0, a.nC, ldCb, kb0, Kb, s + OPB, tid);
    };

    stage(0, 0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) stage(kt + 1, buf ^ 1);
        const char* sR = smem + buf * STAGEB;
        const char* sC = sR + OPB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = 2 * ks + hh;
            uint4 fc[2], fr[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fc[t] = lds_frag(sC, wc * 64 + t * 32 + l31, chunk);
                fr[t] = lds_frag(sR, wr * 64 + t * 32 + l31, chunk);
            }
#pragma unroll
            for (int tr = 0; tr < 2; ++tr)
#pragma unroll
                for (int tc = 0; tc < 2; ++tc) {
                    if constexpr (MODE == GEMM_F32) {
                        const float* pa = (const float*)&fc[tc];
                        const float* pb = (const float*)&fr[tr];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[e], pb[e], acc[tr][tc], 0, 0, 0);
                    } else if constexpr (MODE == GEMM_F16) {
                        acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(f16x8, fc[tc]), __builtin_bit_cast(f16x8, fr[tr]), acc[tr][tc], 0, 0, 0);
                    } else {
                        acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, fc[tc]), __builtin_bit_cast(bf16x8, fr[tr]), acc[tr][tc], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds out[rr][cc..cc+3] for reg quad q of tile (tr,tc)
    const bool vec_ok = a.out && ((a.ldo & 3) == 0) && ((((uintptr_t)a.out) & 15) == 0);
#pragma unroll
    for (int tr = 0; tr < 2; ++tr) {
        const int rr = r0 + wr * 64 + tr * 32 + l31;
        const bool row_ok = rr < a.nR;
        int cnt = 0;
        int gt = -1;
        float sg = 0.0f;
        if (a.count && row_ok) {
            gt = a.gt_col[rr] - a.col0;
            sg = a.s_gt[rr];
        }
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cc = c0 + wc * 64 + tc * 32 + 8 * q + 4 * hh;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[tr][tc][4 * q + e] * a.scale;
                if (a.bias || a.bn_scale || a.act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = cc + e;
                        if (c < a.nC) {
                            float t = v[e];
                            if (a.bias) t += a.bias[c];
                            t = act_apply(t, a.act);
                            if (a.bn_scale) t = t * a.bn_scale[c] + a.bn_shift[c];
                            v[e] = t;
                        }
                    }
                }
                if (a.count && row_ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = cc + e;
                        cnt += (c < a.nC && c != gt && v[e] > sg) ? 1 : 0;
                    }
                }
                if (a.out && row_ok) {
                    float* o = a.out + (long)rr * a.ldo + cc;
                    if (vec_ok && cc + 3 < a.nC) {
                        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (cc + e < a.nC) o[e] = v[e];
                    }
                }
            }
        }
        if (a.count) {
            cnt += __shfl_xor(cnt, 32);
            if (hh == 0 && row_ok && cnt) atomicAdd(a.count + rr, cnt);
        }
    }
}

template <int MODE, bool GLDS>
static hipError_t launch_t(const GemmArgs& a, hipStream_t st) {
    const int tiles_r = (a.nR + TILE - 1) / TILE, tiles_c = (a.nC + TILE - 1) / TILE;
    const long nb = (long)tiles_r * tiles_c;
    if (nb <= 0 || nb > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm_nt_kernel<MODE, GLDS>), dim3((unsigned)nb), dim3(256), 2 * STAGEB, st, a);
    return hipGetLastError();
}

hipError_t launch_gemm_nt(const GemmArgs& a, int mode, bool glds, hipStream_t st) {
    switch (mode) {
        case GEMM_F32: return glds ? launch_t<GEMM_F32, true>(a, st) : launch_t<GEMM_F32, false>(a, st);
        case GEMM_F16: return glds ? launch_t<GEMM_F16, true>(a, st) : launch_t<GEMM_F16, false>(a, st);
        case GEMM_BF16: return glds ? launch_t<GEMM_BF16, true>(a, st) : launch_t<GEMM_BF16, false>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace laff
