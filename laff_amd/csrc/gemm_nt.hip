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
// Structure: WR x WC waves per workgroup, each wave owns (WM*32) x (WN*32) outputs as WM x WN MFMA tiles of 32x32.
// K-step = 128 bytes per row (64 halves / 32 floats); two LDS stages; operands staged global -> LDS with 16-byte
// direct-to-LDS loads (global_load_lds_dwordx4) issued from inline asm with hand-counted vmcnt (hipcc would otherwise
// put `s_waitcnt vmcnt(0)` in front of every K-step's first ds_read).  The LDS image is lane-linear, so the
// bank-conflict swizzle is applied to the per-lane SOURCE address and undone on the ds_read_b128 side
// (chunk ^= (row>>1)&7 inside each 128-byte row: SQ_LDS_BANK_CONFLICT = 0 measured).
// The MFMA "A" operand is fed from C rows and "B" from R rows so that each lane ends up with 4 consecutive output
// columns of one output row per accumulator quad.
//
// Three tile bodies are instantiated:
//   Cfg128 : 2x2 waves of 64x64   -> 128x128 tile, 256 threads, 64 KiB LDS, 2 workgroups / CU   (fp32 FC, small problems)
//   Cfg256 : 2x4 waves of 128x64  -> 256x256 tile, 512 threads, 128 KiB LDS, 1 workgroup / CU   (16-bit similarity:
//            half the operand bytes per flop through the per-CU load path, 3/4 of the LDS reads per MFMA)
//   CfgX3  : the Cfg256 wave layout on 32-element K-steps holding all four planes of a hi/lo split product, the three
//            partial products interleaved per 16-element slice (gemm_tile_x3: split FC, fp16x3 / bf16x3 similarity)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "kernels.h"

namespace laff {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0, 0, 0, 0};

constexpr int ROWB = 128;   // bytes of K per row per K-step

// Similarity kernels only: staging lists for the pairs inside the error band, placed BEYOND the operand ring (the ring is still
// being read when the first waves reach the epilogue): one list of PAIR_LCAP / waves {row, col} pairs per wave.  (Appending every
// pair with its own atomic on the one global counter serialised the whole launch: 92k pairs at C4 x ~12 ns = 1.1 ms for a 0.6 ms
// kernel; a returning atomic per tile still cost its round trip at the very end of every tile.)
constexpr int PAIR_LCAP = 1024;
// ... followed by the per-row inputs of the banded epilogue for the tile's (at most 256) rows, brought in by LDS-DMA next to the first
// operand stage: gt_col [256] int32 | band_r [256] fp32 | s_gt64 [256] fp64
constexpr int ROWDATA_OFF = 16 + PAIR_LCAP * 8;
constexpr int ROWDATA_GT = 0, ROWDATA_BR = 1024, ROWDATA_SG = 2048, ROWDATA_BC = 4096, ROWDATA_BYTES = 5120;
// (ROWDATA_BC: the 16 bytes of per-64-column band maxima that cover the tile's columns, replicated by every lane of one piece)
constexpr int PAIR_LDS = ROWDATA_OFF + ROWDATA_BYTES;
// slots per tile segment: half of the list is split evenly between the tiles, the other half takes the overflow
__host__ __device__ __forceinline__ unsigned pair_chunk(unsigned pair_cap, unsigned ntiles) {
    unsigned c = (pair_cap / 2u) / (ntiles ? ntiles : 1u);
    c = c > (unsigned)PAIR_LCAP ? (unsigned)PAIR_LCAP : c;
    return c & ~3u;                                                         // 0: everything goes through the counter
}

template <int WM_, int WN_, int WR_, int WC_>
struct Cfg {
    static constexpr int WM = WM_, WN = WN_, WR = WR_, WC = WC_;
    static constexpr int TR = WR * WM * 32, TC = WC * WN * 32;     // output tile
    static constexpr int THREADS = 64 * WR * WC;
    static constexpr int OPB_R = TR * ROWB, OPB_C = TC * ROWB, STAGEB = OPB_R + OPB_C;
    static constexpr int SMEM = 2 * STAGEB;
    static constexpr int ITR = TR * 8 / THREADS, ITC = TC * 8 / THREADS;   // 16-byte chunks per thread per stage
    static constexpr int WPS = (THREADS / 64) * ((160 * 1024) / (SMEM + PAIR_LDS)) / 4;  // waves per SIMD the LDS budget admits
    static constexpr int PITCH = WN * 32 + 4;                        // epilogue slab: 32 rows x (WN*32) fp32 per wave
    static_assert(WN == 2 || WN == 4, "the epilogue store loop handles 64 or 128 output columns per wave");
    static_assert(THREADS / 64 * 32 * PITCH * 4 <= SMEM, "epilogue slabs must fit in the operand ring");
};
using Cfg128 = Cfg<2, 2, 2, 2>;
using Cfg256 = Cfg<4, 2, 2, 4>;
using Cfg256L = Cfg<4, 4, 2, 2>;     // the same tile on 4 waves of 128x128: one 512-register wave per SIMD, for long K

template <int MODE> struct ModeTraits;
template <> struct ModeTraits<GEMM_F32> { static constexpr int ESZ = 4; };
template <> struct ModeTraits<GEMM_F16> { static constexpr int ESZ = 2; };
template <> struct ModeTraits<GEMM_BF16> { static constexpr int ESZ = 2; };

// tanh / sigmoid on the hardware exp2 + rcp (v_exp_f32, v_rcp_f32: ~1 ulp each): |error| <= ~2.4e-7 absolute (measured
// against tanhf over [-12, 12]).  ocml's tanhf costs ~100 instructions per value, which at 64 outputs per lane is as
// long as the tile's whole fp32 MFMA stream.
__device__ __forceinline__ float fast_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);     // e^(2x); inf / 0 saturate correctly
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Lanes of one wavefront exchanging data through LDS: the hardware executes a wavefront's LDS operations in order, so all that is
// needed is that the COMPILER keeps the stores in front of the loads (and the loads in front of the next stores).
// __builtin_amdgcn_wave_barrier() alone is declared without memory effects, so per-thread alias analysis (a lane's own store and
// load addresses differ) would be free to move one across the other; the fences make the ordering explicit.
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    asm volatile("" ::: "memory");
}

// ---- staging, generic paths (STG 0: through registers, element-wise K bounds; STG 1: LDS-DMA with per-step address
// arithmetic, 16-byte K granularity).  One operand tile = ROWS x 8 chunks of 16 B; LDS slot p = row*8 + cs holds
// source chunk swz(row, cs).
template <int STG, int ROWS, int THREADS>
__device__ __forceinline__ void stage_operand(const char* __restrict__ base, int row0, int nrows, long ldb, long kbyte0,
                                              long kbytes_valid, char* lds_op, unsigned lds_op_addr, int tid) {
#pragma unroll
    for (int it = 0; it < ROWS * 8 / THREADS; ++it) {
        const int p = it * THREADS + tid;
        const int row = p >> 3, cs = p & 7;
        const int c = swz(row, cs);
        int gr = row0 + row;
        gr = gr < nrows ? gr : nrows - 1;        // clamp: garbage rows are never stored
        const long kb = kbyte0 + (long)c * 16;
        if constexpr (STG == 1) {
            const char* src = (kb + 16 <= kbytes_valid) ? base + (long)gr * ldb + kb : (const char*)g_zero16;
            const unsigned dst = lds_op_addr + (unsigned)(it * THREADS + (tid & ~63)) * 16u;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, off\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst))
                : "memory");
        } else {
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
    return *(const uint4*)(lds_op + row * ROWB + swz(row, chunk) * 16);
}

// ---- staging, fast path (STG 2: K bytes a multiple of the K-step, operand below 4 GiB): the per-lane part of every
// source address is a 32-bit byte offset computed ONCE per tile; a K-step only advances a scalar base (saddr form), so
// the main loop carries no address VALU.  LDS-DMA lands 16 B per lane at (wave-uniform M0 base) + lane*16.
// one 1 KiB-per-wave piece of a stage (the K loop spreads a stage's pieces between MFMAs instead of issuing them in a burst)
__device__ __forceinline__ void glds_piece(unsigned off, unsigned long long sbase, unsigned dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(off), "s"(sbase), "s"(dst)
        : "memory");
}


// the same with the LDS base given as scalar base + constant and M0 LEFT pointing at the piece: 3 instructions instead of 6 (every
// issue slot saved in the K loop is an MFMA-pipe bubble less).  M0 is a reserved register the compiler does not track through an
// asm statement, so the K loops that use this bracket themselves with m0_save() / m0_restore(); nothing the compiler generates
// inside them reads M0 (their only LDS-DMA is this helper).
template <int IMM>
__device__ __forceinline__ void glds_piece_s(unsigned off, unsigned long long sbase, unsigned dst_base) {
    asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(off), "s"(sbase), "s"(dst_base), "n"(IMM)
                 : "memory", "scc");          // s_add_i32 writes SCC (a carry chain of the compiler's may be in flight around the asm)
}
__device__ __forceinline__ unsigned m0_save() {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep)::"memory");
    return keep;
}
__device__ __forceinline__ void m0_restore(unsigned keep) { asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory"); }

__device__ __forceinline__ unsigned long long uniform64(unsigned long long x) {     // make wave-uniformity provable
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// blockIdx -> linear tile id such that each XCD (block b runs on XCD b % 8) works on a contiguous chunk of tiles
// and neighbouring tiles share operand panels in that XCD's L2 (bijective for any nb).
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- epilogue ----------------------------------------------------------------------------------------------------
// Accumulator layout: lane (l31, hh) holds out[row l31 of the 32-row block][4 consecutive columns] per register quad.
// The per-column epilogue runs in that layout; the block then goes through a wave-private LDS slab (32 x 64 fp32, row
// pitch 68 words) so that every global store instruction writes 4 rows x 256 contiguous bytes instead of 64 scattered
// 16-byte pieces.  EPI_SIM: scale (+ ground-truth patch and rank count); EPI_FC: row/column scales, bias, activation,
// folded BatchNorm.  FULL = interior tile with 16-byte aligned output: no bounds logic at all (a first version with
// per-element bounds and per-element activation switches compiled to 12k lines of branches and cost 37 % of the tile).
enum { EPI_SIM = 0, EPI_FC = 1 };

template <int EPI, bool FULL, typename CF>
__device__ __forceinline__ void epilogue(const GemmArgs& a, f32x16 (&acc)[CF::WM][CF::WN], int r0, int c0, int wr, int wc,
                                         int wave, int lane, char* smem) {
    constexpr int WM = CF::WM, WN = CF::WN, PITCH = CF::PITCH;
    const int l31 = lane & 31, hh = lane >> 5;
    float* slab = (float*)smem + wave * (32 * PITCH);
    const int cw0 = c0 + wc * (WN * 32);               // first output column of this wave
    const bool counting = EPI == EPI_SIM && a.count != nullptr;
#pragma unroll
    for (int tr = 0; tr < WM; ++tr) {
        const int rbase = r0 + wr * (WM * 32) + tr * 32;
        const int rr = rbase + l31;
        const bool row_ok = FULL || rr < a.nR;
        int cnt = 0;
        int gt = -1;
        float sg = 0.0f;
        if (counting && row_ok) {
            gt = a.gt_col[rr] - a.col0;
            sg = a.s_gt[rr];
        }
        float rscl = a.scale;
        if (EPI == EPI_FC && a.row_scale && row_ok) rscl *= a.row_scale[rr];
#pragma unroll
        for (int tc = 0; tc < WN; ++tc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = tc * 32 + 8 * q + 4 * hh;          // column inside the wave's 64
                const int cc = cw0 + cl;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[tr][tc][4 * q + e] * rscl;
                if constexpr (EPI == EPI_FC) {
                    float cs[4] = {1, 1, 1, 1}, bb[4] = {0, 0, 0, 0}, ss[4] = {1, 1, 1, 1}, hs[4] = {0, 0, 0, 0};
                    if (FULL || cc + 3 < a.nC) {          // per-column parameters as 16-byte loads
                        if (a.col_scale) *(float4*)cs = *(const float4*)(a.col_scale + cc);
                        if (a.bias) *(float4*)bb = *(const float4*)(a.bias + cc);
                        if (a.bn_scale) {
                            *(float4*)ss = *(const float4*)(a.bn_scale + cc);
                            *(float4*)hs = *(const float4*)(a.bn_shift + cc);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (cc + e < a.nC) {
                                if (a.col_scale) cs[e] = a.col_scale[cc + e];
                                if (a.bias) bb[e] = a.bias[cc + e];
                                if (a.bn_scale) { ss[e] = a.bn_scale[cc + e]; hs[e] = a.bn_shift[cc + e]; }
                            }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], cs[e], bb[e]);
                    if (a.act == 1) {                      // wave-uniform: one scalar branch per quad
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fast_tanh(v[e]);
                    } else if (a.act == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                    } else if (a.act == 3) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fast_sigmoid(v[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], ss[e], hs[e]);
                }
                if (counting) {
                    // the ground-truth entry is DEFINED by the pre-pass value s_gt (laff_row_dot_gt); writing it into S keeps
                    // "rank counted here" == "rank recounted from S" exactly (and excludes it from the count: sg > sg is false)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (cc + e == gt) ? sg : v[e];
                        const bool in = FULL || (row_ok && cc + e < a.nC);
                        cnt += (in && v[e] > sg) ? 1 : 0;
                    }
                }
                if (a.out) *(float4*)(slab + l31 * PITCH + cl) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if (counting) {
            cnt += __shfl_xor(cnt, 32);
            if (hh == 0 && row_ok && cnt) atomicAdd(a.count + rr, cnt);
        }
        if (a.out) {
            wave_lds_fence();
            constexpr int LPR = WN * 8, RPI = 64 / LPR;      // lanes per slab row, rows per store instruction
            const int col4 = (lane % LPR) * 4;
            const int gc = cw0 + col4;
#pragma unroll
            for (int j = 0; j < 32 / RPI; ++j) {
                const int row = lane / LPR + RPI * j;
                const float4 v = *(const float4*)(slab + row * PITCH + col4);
                const int gr = rbase + row;
                float* o = a.out + (long)gr * a.ldo + gc;
                if constexpr (FULL) {
                    // streamed once, never re-read by this kernel: keep the score tile out of the L2 that holds the panels
                    __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), (f32x4*)o);
                } else if (gr < a.nR) {
                    const bool vec_ok = ((a.ldo & 3) == 0) && ((((uintptr_t)a.out) & 15) == 0);
                    if (vec_ok && gc + 3 < a.nC) {
                        *(float4*)o = v;
                    } else {
                        const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (gc + e < a.nC) o[e] = t[e];
                    }
                }
            }
            wave_lds_fence();                           // slab is rewritten by the next 32-row block
        }
    }
}

// ---- FC epilogue: row/column scales of the operand split, bias, activation, folded BatchNorm ---------------------------------
// The per-column parameters (col_scale, bias, bn_scale, bn_shift: 4 x 16 bytes per quad of columns) depend on the column only.  The
// generic epilogue above fetched them inside the (row block, column block, quad) loop, behind the previous block's stores (the
// compiler cannot hoist a load across a store through an unrelated pointer): 32 dependent L2 round trips per lane -- the epilogue
// took 25-31k cycles of a 112k-cycle fused-split tile (tools/debug/trace_fc.py), three times the similarity GEMM's.  Here the
// column block is the OUTER loop: its 16 parameter vectors are loaded once into registers (64 VGPRs: the K loop's fragment registers
// are dead by now) and reused for all WM row blocks; each (row block, column block) goes through a 32 x 32 slab, every store
// instruction writes 8 rows x 128 contiguous bytes (whole cache lines: tile columns are multiples of 32).
template <bool FULL, typename CF>
__device__ __forceinline__ void epilogue_fc(const GemmArgs& a, f32x16 (&acc)[CF::WM][CF::WN], int r0, int c0, int wr, int wc,
                                            int wave, int lane, char* smem) {
    constexpr int WM = CF::WM, WN = CF::WN, P32 = 36;                       // slab pitch in words: 32 columns + 4
    const int l31 = lane & 31, hh = lane >> 5;
    float* slab = (float*)smem + wave * (32 * P32);
    const int cw0 = c0 + wc * (WN * 32);
    float rscl[WM];
#pragma unroll
    for (int tr = 0; tr < WM; ++tr) {
        const int rr = r0 + wr * (WM * 32) + tr * 32 + l31;
        rscl[tr] = a.scale;
        if (a.row_scale && (FULL || rr < a.nR)) rscl[tr] *= a.row_scale[rr];
    }
#pragma unroll
    for (int tc = 0; tc < WN; ++tc) {
        float cs[4][4], bb[4][4], ss[4][4], hs[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cc = cw0 + tc * 32 + 8 * q + 4 * hh;
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[q][e] = 1.0f; bb[q][e] = 0.0f; ss[q][e] = 1.0f; hs[q][e] = 0.0f; }
            if (FULL || cc + 3 < a.nC) {
                auto ld4 = [](const float* p, float (&o)[4]) {
                    const float4 t = *(const float4*)p;
                    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
                };
                if (a.col_scale) ld4(a.col_scale + cc, cs[q]);
                if (a.bias) ld4(a.bias + cc, bb[q]);
                if (a.bn_scale) {
                    ld4(a.bn_scale + cc, ss[q]);
                    ld4(a.bn_shift + cc, hs[q]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (cc + e < a.nC) {
                        if (a.col_scale) cs[q][e] = a.col_scale[cc + e];
                        if (a.bias) bb[q][e] = a.bias[cc + e];
                        if (a.bn_scale) { ss[q][e] = a.bn_scale[cc + e]; hs[q][e] = a.bn_shift[cc + e]; }
                    }
            }
        }
#pragma unroll
        for (int tr = 0; tr < WM; ++tr) {
            const int rbase = r0 + wr * (WM * 32) + tr * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[tr][tc][4 * q + e] * rscl[tr];
                // The four products are pinned in registers before the multiply-adds.  Written as one expression, the compiler emits
                //   v_pk_mul_f32 P, acc, rscl ; s_nop 0 ; v_pk_fma_f32 .., P, .. ; v_pk_mul_f32 P, .. ; s_nop 0 ; v_pk_fma_f32 .., P, ..
                // directly behind the previous block's four 16-byte stores, and on MI355X that sequence sporadically delivered a STALE
                // P to the second multiply-add in lanes 48..63 (measured: ~20 events of 16 rows x 1 column per 416-tile launch, only in
                // workgroups that are not the first on their CU, only in the first quad of row blocks 1 and 3; any change of the
                // sequence -- this pin, plain v_mul_f32, even moving the slab 64 KiB up in LDS -- gave 0 events in 30 x 416 tiles,
                // tools/debug/stress_fc.py).  Cause not established; the pin costs two VALU issue slots per quad.
                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], cs[q][e], bb[q][e]);
                if (a.act == 1) {                          // wave-uniform
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_tanh(v[e]);
                } else if (a.act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                } else if (a.act == 3) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_sigmoid(v[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], ss[q][e], hs[q][e]);
                *(float4*)(slab + l31 * P32 + 8 * q + 4 * hh) = make_float4(v[0], v[1], v[2], v[3]);
            }
            wave_lds_fence();
            const int col4 = (lane & 7) * 4;
            const int gc = cw0 + tc * 32 + col4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (lane >> 3) + 8 * j;
                const float4 v = *(const float4*)(slab + row * P32 + col4);
                const int gr = rbase + row;
                float* o = a.out + (long)gr * a.ldo + gc;
                if constexpr (FULL) {
                    __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), (f32x4*)o);     // keeps the operand panels in L2
                } else if (gr < a.nR) {
                    const bool vec_ok = ((a.ldo & 3) == 0) && ((((uintptr_t)a.out) & 15) == 0);
                    if (vec_ok && gc + 3 < a.nC) {
                        *(float4*)o = v;
                    } else {
                        const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (gc + e < a.nC) o[e] = t[e];
                    }
                }
            }
            wave_lds_fence();                           // the slab is rewritten by the next block
        }
    }
}

// ---- exact-rank ("banded") epilogue of the similarity GEMM ------------------------------------------------------------------
// count[row] += #{col != gt : S > s_gt + band}; the pairs with |S - s_gt| <= band are staged for laff_rank_resolve (rank.hip), which
// re-scores them exactly; S (optional) gets the exact ground-truth score at the ground-truth entry.  band = band_r[row] + the
// maximum of band_c over the wave's 64 columns (computed once per 64-column block by laff_rank_prepare: one value per wave).
//   * The per-row inputs (gt_col, s_gt64, band_r of the tile's rows) arrive in LDS by DMA next to the first operand stage
//     (band_stage_rows): an L2 round trip per 32-row block in front of the stores was ~1/4 of this epilogue.
//   * The tests run in ACCUMULATOR units -- thresholds lo = (s_gt - band) / scale, hi = (s_gt + band) / scale per row -- and with
//     vector registers only: two per-lane counters (x > hi, x >= lo), i.e. two compares and two add-with-carry per element and no
//     VALU -> SGPR -> SALU hazard stalls; a 32x32 block holds a pair in the band iff the two counters moved apart: ONE lane-mask
//     test per block guards the rare part.
//   * The rare part turns the lane's in-band elements into a bit mask and compacts them wave-wide (ballot + mbcnt) into a
//     wave-private LDS list: no LDS atomics, no workgroup barrier.
//   * Each WAVE publishes its list into its own fixed segment of the global list (unused slots marked invalid): no global atomic.
//     Only a wave with more pairs than its segment appends the excess behind the segments with the global counter.
// Row inputs of the banded epilogue -> LDS, by LDS-DMA: waves 0..4 issue ONE extra 1 KiB piece each (gt_col, band_r, two halves of
// s_gt64 for a 256-row tile, the column-block band maxima) right behind their pieces of the first operand stage.  Being the youngest requests of those waves they
// are left in flight by the prologue's wait (`vmcnt(1)`) and are covered by the `vmcnt(0)` + barrier of the first K-step: no
// registers, no address arithmetic, no wait of their own.  (As per-lane register loads the same 13 requests -- cold lines written
// by laff_rank_prepare on other XCDs -- cost 1.7k cycles per tile wherever they were issued: in front of the first barrier, behind
// it, or in the middle of the K loop with a counted wait.)  A lane fetches the aligned 16 bytes that hold its rows; groups beyond
// the last row are redirected to the last valid group (their rows are never used), so the arrays must be 16-byte aligned and
// readable up to the next multiple of 16 bytes (any hipMalloc / torch allocation is).
template <typename CF>
__device__ __forceinline__ void band_stage_piece(const GemmArgs& a, int r0, int c0, int wave, int lane, unsigned lds0);

template <typename CF>
__device__ __forceinline__ int band_stage_rows(const GemmArgs& a, int r0, int c0, int wave, int lane, unsigned lds0) {
    constexpr int TR = CF::TR;
    const bool on = a.s_gt64 != nullptr && a.count != nullptr;
    // piece w of waves 0..3: 0 = gt_col (4 rows per lane), 1 = band_r (4 rows per lane), 2/3 = s_gt64 rows [0,128) / [128,256) (2 per lane)
    constexpr int NP = TR > 128 ? 4 : 3, NWAVES = CF::THREADS / 64;
    if (!on) return 0;
    if constexpr (NWAVES <= NP) {
        // fewer waves than pieces (4-wave configurations): wave 0 also issues the piece a fifth wave would have
        static_assert(NWAVES == NP, "row inputs: one piece per wave, the last one doubled up on wave 0");
        if (wave == 0) band_stage_piece<CF>(a, r0, c0, NP, lane, lds0);
        band_stage_piece<CF>(a, r0, c0, wave, lane, lds0);
        return wave == 0 ? 2 : 1;
    } else {
        if (wave > NP) return 0;
        band_stage_piece<CF>(a, r0, c0, wave, lane, lds0);
        return 1;
    }
}

template <typename CF>
__device__ __forceinline__ void band_stage_piece(const GemmArgs& a, int r0, int c0, int wave, int lane, unsigned lds0) {
    constexpr int TR = CF::TR;
    constexpr int NP = TR > 128 ? 4 : 3;
    const unsigned dst = lds0 + (unsigned)CF::SMEM + (unsigned)ROWDATA_OFF;
    const int last = a.nR - 1;
    if (wave == NP) {
        // column-block band maxima: laff_rank_prepare stores them behind the per-column values at the 16-byte aligned offset
        // (nC + 3) & ~3; every lane fetches the aligned group of 4 that holds the tile's first block
        const int nblk = (a.nC + 63) >> 6;
        const int g = min((c0 >> 6) & ~3, (nblk - 1) & ~3);
        const unsigned long long base = uniform64((unsigned long long)(const void*)(a.band_c + ((a.nC + 3) & ~3)));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"((unsigned)g * 4u), "s"(base), "s"(__builtin_amdgcn_readfirstlane(dst + ROWDATA_BC))
                     : "memory");
        return;
    }
    if (wave < 2) {
        const int g = min(r0 + 4 * lane, last & ~3);                      // first row of this lane's group of 4
        const unsigned long long base = uniform64((unsigned long long)(wave == 0 ? (const void*)a.gt_col : (const void*)a.band_r));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"((unsigned)g * 4u), "s"(base), "s"(__builtin_amdgcn_readfirstlane(dst + (wave == 0 ? ROWDATA_GT : ROWDATA_BR)))
                     : "memory");
    } else {
        const int half = wave - 2;
        const int g = min(r0 + 128 * half + 2 * lane, last & ~1);         // first row of this lane's pair
        const unsigned long long base = uniform64((unsigned long long)(const void*)a.s_gt64);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"((unsigned)g * 8u), "s"(base), "s"(__builtin_amdgcn_readfirstlane(dst + ROWDATA_SG + 1024 * half))
                     : "memory");
    }
}

template <bool FULL, typename CF>
__device__ __forceinline__ void epilogue_banded(const GemmArgs& a, f32x16 (&acc)[CF::WM][CF::WN], int r0, int c0, int wr, int wc,
                                                int wave, int lane, char* smem) {
    constexpr int WM = CF::WM, WN = CF::WN, PITCH = CF::PITCH, NWAVES = CF::THREADS / 64;
    constexpr unsigned WCAP = PAIR_LCAP / NWAVES;                           // staged pairs per wave
    const int l31 = lane & 31, hh = lane >> 5;
    float* slab = (float*)smem + wave * (32 * PITCH);
    const int cw0 = c0 + wc * (WN * 32);
    unsigned* wl = (unsigned*)(smem + CF::SMEM) + 4 + wave * (2 * WCAP);    // this wave's staging list
    const unsigned wchunk = pair_chunk(a.pair_cap, gridDim.x * NWAVES);     // slots of this wave's segment of the global list
    const size_t ovf_base = (size_t)gridDim.x * NWAVES * wchunk;            // the overflow region starts behind the segments
    unsigned wcount = 0;                                                    // wave-uniform: pairs staged so far
#ifdef LAFF_GEMM_TRACE
#define ETRACE(i) do { if (a.trace && threadIdx.x == 0) a.trace[(long)gridDim.x * 16 + (long)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define ETRACE(i) do {} while (0)
#endif
    ETRACE(0);
    const float inv_scale = 1.0f / a.scale;
    // the wave's column-block band (one wave-uniform value) and the lane's rows from the LDS copy the prologue brought in
    const char* rowdata = smem + CF::SMEM + ROWDATA_OFF;
    float bc = 0.0f;
    {
        const int nblk = (a.nC + 63) >> 6;
        const int g0 = min((c0 >> 6) & ~3, (nblk - 1) & ~3);     // first block of the staged group of 4 (see band_stage_rows)
#pragma unroll
        for (int i = 0; i < WN / 2; ++i) {
            const int c = cw0 + i * 64;                         // wave-uniform
            if (c < a.nC) bc = fmaxf(bc, ((const float*)(rowdata + ROWDATA_BC))[(c >> 6) - g0]);
        }
    }
    int gt_[WM];
    float br_[WM];
    double sg_[WM];
#pragma unroll
    for (int tr = 0; tr < WM; ++tr) {
        const int rt = wr * (WM * 32) + tr * 32 + l31;          // row inside the tile
        gt_[tr] = ((const int*)(rowdata + ROWDATA_GT))[rt] - a.col0;
        br_[tr] = ((const float*)(rowdata + ROWDATA_BR))[rt];
        sg_[tr] = ((const double*)(rowdata + ROWDATA_SG))[rt];
    }
#pragma unroll
    for (int tr = 0; tr < WM; ++tr) {
        const int rbase = r0 + wr * (WM * 32) + tr * 32;
        const int rr = rbase + l31;
        const bool row_ok = FULL || rr < a.nR;
        ETRACE(1 + tr);
        const int gt = row_ok ? gt_[tr] : -1;
        const float sg = (float)sg_[tr], eps = br_[tr] + bc;
        // accumulator-unit thresholds (the band's constant term carries the rounding of these two products)
        const float lo = (sg - eps) * inv_scale, hi = (sg + eps) * inv_scale;
        int c_hi = 0, c_lo = 0;                                             // #{x > hi}, #{x >= lo} over this lane's elements
#pragma unroll
        for (int tc = 0; tc < WN; ++tc) {
            const int cb = cw0 + tc * 32;
            const int before = c_lo - c_hi;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float x = acc[tr][tc][i];
                if constexpr (FULL) {
                    c_hi += (x > hi) ? 1 : 0;
                    c_lo += (x >= lo) ? 1 : 0;
                } else {
                    const bool in = row_ok && cb + 8 * (i >> 2) + 4 * hh + (i & 3) < a.nC;
                    c_hi += (in && x > hi) ? 1 : 0;
                    c_lo += (in && x >= lo) ? 1 : 0;
                }
            }
            const bool gt_here = (unsigned)(gt - cb) < 32u;                  // the ground-truth column is in this block
            int gi = -1;                                                     // ... and this lane holds it, as element gi
            if (__builtin_amdgcn_ballot_w64(gt_here || (c_lo - c_hi) != before) != 0ull) {      // ~1/3 of the blocks
                unsigned bits = 0u;                                          // element i in the band [lo, hi] <-> bit i
#pragma unroll
                for (int i = 15; i >= 0; --i) {
                    const float x = acc[tr][tc][i];
                    const bool in = FULL || (row_ok && cb + 8 * (i >> 2) + 4 * hh + (i & 3) < a.nC);
                    // med3(x, lo, hi) == x  <=>  lo <= x <= hi (false for NaN): one compare, no scalar mask arithmetic
                    bits = bits + bits + ((in && __builtin_amdgcn_fmed3f(x, lo, hi) == x) ? 1u : 0u);
                }
                if (__builtin_amdgcn_ballot_w64(gt_here) != 0ull) {          // scalar branch, ~1/10 of the blocks
                    const int j = gt - cb;
                    const bool mine = gt_here && ((j >> 2) & 1) == hh && row_ok && (FULL || gt < a.nC);
                    gi = mine ? (j >> 3) * 4 + (j & 3) : -1;
#pragma unroll
                    for (int i = 0; i < 16; ++i) c_hi -= (i == gi && acc[tr][tc][i] > hi) ? 1 : 0;    // never counted
                    if (mine) bits &= ~(1u << gi);                           // never listed
                }
                unsigned long long m;
                while ((m = __builtin_amdgcn_ballot_w64(bits != 0u)) != 0ull) {     // one trip per pair of the busiest lane
                    if (bits != 0u) {
                        const int i = __builtin_ctz(bits);
                        bits &= bits - 1u;
                        const unsigned cc = (unsigned)(cb + 8 * (i >> 2) + 4 * hh + (i & 3));
                        const unsigned slot = wcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (slot < WCAP) {
                            wl[2 * slot] = (unsigned)rr;
                            wl[2 * slot + 1] = cc;
                        } else {                                             // staging list full: straight to the global list
                            const size_t g = ovf_base + atomicAdd(a.pairs, 1u);
                            if (g < a.pair_cap) {
                                a.pairs[4 + 2 * g] = (unsigned)rr;
                                a.pairs[5 + 2 * g] = cc;
                            }
                        }
                    }
                    wcount += (unsigned)__builtin_popcountll(m);
                }
            }
            if (a.out) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[tr][tc][4 * q + e] * a.scale;
                    *(float4*)(slab + l31 * PITCH + tc * 32 + 8 * q + 4 * hh) = make_float4(v[0], v[1], v[2], v[3]);
                }
                // the exact score at the ground-truth entry: one lane of one block per row patches its slab word (a per-element select
                // here was two of the epilogue's six vector instructions per accumulator, and the epilogue is VALU-bound)
                if (gi >= 0) slab[l31 * PITCH + tc * 32 + 8 * (gi >> 2) + 4 * hh + (gi & 3)] = sg;
            }
        }
        int cnt = c_hi + __shfl_xor(c_hi, 32);
        if (hh == 0 && row_ok && cnt) atomicAdd(a.count + rr, cnt);
        if (a.out) {
            wave_lds_fence();
            constexpr int LPR = WN * 8, RPI = 64 / LPR;      // lanes per slab row, rows per store instruction
            const int col4 = (lane % LPR) * 4;
            const int gc = cw0 + col4;
#pragma unroll
            for (int j = 0; j < 32 / RPI; ++j) {
                const int row = lane / LPR + RPI * j;
                const float4 v = *(const float4*)(slab + row * PITCH + col4);
                const int gr = rbase + row;
                float* o = a.out + (long)gr * a.ldo + gc;
                if constexpr (FULL) {
                    __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), (f32x4*)o);
                } else if (gr < a.nR) {
                    const bool vec_ok = ((a.ldo & 3) == 0) && ((((uintptr_t)a.out) & 15) == 0);
                    if (vec_ok && gc + 3 < a.nC) {
                        *(float4*)o = v;
                    } else {
                        const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (gc + e < a.nC) o[e] = t[e];
                    }
                }
            }
            wave_lds_fence();                           // slab is rewritten by the next 32-row block
        }
    }
    ETRACE(5);
    // publish: this wave's segment [w * wchunk, (w + 1) * wchunk) of the global list, valid pairs first, the rest marked invalid
    wcount = __builtin_amdgcn_readfirstlane(wcount);
    const unsigned n = min(wcount, WCAP);
    const size_t seg = ((size_t)blockIdx.x * NWAVES + wave) * wchunk;
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.pairs[2] = gridDim.x * NWAVES * wchunk; a.pairs[3] = wchunk; }
    for (unsigned i = lane; i < wchunk; i += 64) {
        const bool have = i < n;
        a.pairs[4 + 2 * (seg + i)] = have ? wl[2 * i] : 0xffffffffu;
        a.pairs[5 + 2 * (seg + i)] = have ? wl[2 * i + 1] : 0u;
    }
    ETRACE(6);
    if (n > wchunk) {                                                       // wave-uniform, rare
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(a.pairs, n - wchunk);
        base = __builtin_amdgcn_readfirstlane(base);
        for (unsigned i = wchunk + lane; i < n; i += 64)
            if (ovf_base + base + (i - wchunk) < a.pair_cap) {
                a.pairs[4 + 2 * (ovf_base + base + (i - wchunk))] = wl[2 * i];
                a.pairs[5 + 2 * (ovf_base + base + (i - wchunk))] = wl[2 * i + 1];
            }
    }
    ETRACE(7);
#undef ETRACE
}

// ---- one output tile ----------------------------------------------------------------------------------------------
// linear tile id -> output origin: groups of 8 tile rows swept along c, so 8 R panels + 8 C panels live in L2 at a time
template <typename CF>
__device__ __forceinline__ void tile_origin(const GemmArgs& a, const int lin, int& r0, int& c0) {
    const int tiles_r = (a.nR + CF::TR - 1) / CF::TR, tiles_c = (a.nC + CF::TC - 1) / CF::TC;
    const int gsz_full = 8 * tiles_c;
    const int grp = lin / gsz_full;
    const int first_r = grp * 8;
    const int gsz = min(tiles_r - first_r, 8);
    const int in_grp = lin - grp * gsz_full;
    r0 = (first_r + in_grp % gsz) * CF::TR;
    c0 = (in_grp / gsz) * CF::TC;
}

template <int MODE, int STG, typename CF, int EPI>
__device__ __forceinline__ void gemm_tile(const GemmArgs& a, const int r0, const int c0, char* smem) {
    constexpr int ESZ = ModeTraits<MODE>::ESZ;
    constexpr int WM = CF::WM, WN = CF::WN, THREADS = CF::THREADS;
    constexpr bool GLDS = STG != 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / CF::WC, wc = wave % CF::WC;
    const int l31 = lane & 31, hh = lane >> 5;


    const long ldRb = (long)a.ldR * ESZ, ldCb = (long)a.ldC * ESZ;
    const long Kb = (long)a.K * ESZ;
    const int kt_per_seg = (int)((Kb + ROWB - 1) / ROWB);
    const int nkt = kt_per_seg * a.nseg;

#ifdef LAFF_GEMM_TRACE
#define TRACE(i) do { if (a.trace && tid == 0) a.trace[(long)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    if (a.trace && tid == 0)
        a.trace[(long)blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xf814) /*XCC_ID*/ << 32) |
                                           __builtin_amdgcn_s_getreg(0xf804) /*HW_ID*/;
#else
#define TRACE(i) do {} while (0)
#endif
    TRACE(0);
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    // per-lane source offsets of this tile (fast path): slot p = it*THREADS + tid -> (row, swizzled chunk)
    unsigned offR[CF::ITR], offC[CF::ITC];
    if constexpr (STG == 2) {
#pragma unroll
        for (int it = 0; it < CF::ITR; ++it) {
            const int p = it * THREADS + tid, row = p >> 3;
            offR[it] = (unsigned)min(r0 + row, a.nR - 1) * (unsigned)ldRb + (unsigned)swz(row, p & 7) * 16u;
        }
#pragma unroll
        for (int it = 0; it < CF::ITC; ++it) {
            const int p = it * THREADS + tid, row = p >> 3;
            offC[it] = (unsigned)min(c0 + row, a.nC - 1) * (unsigned)ldCb + (unsigned)swz(row, p & 7) * 16u;
        }
    }
    // Running scalar state of the operand stream (segment, K-step inside it, current base addresses): advanced by one
    // K-step per stage_next() call, so the loop needs no division, no table lookup and no scalar memory load (scalar
    // loads share lgkmcnt with the hand-counted LDS reads below and return out of order).
    int st_seg = 0, st_kin = 0;
    unsigned long long curR = (unsigned long long)((const char*)a.R + a.segR[0]);
    unsigned long long curC = (unsigned long long)((const char*)a.C + a.segC[0]);
    const unsigned long long nR1 = (unsigned long long)((const char*)a.R + a.segR[1]), nC1 = (unsigned long long)((const char*)a.C + a.segC[1]);
    const unsigned long long nR2 = (unsigned long long)((const char*)a.R + a.segR[2]), nC2 = (unsigned long long)((const char*)a.C + a.segC[2]);
    auto stage_next = [&](int buf) {             // stages K-steps in order: call k stages step k
        const long kb0 = (long)st_kin * ROWB;
        char* s = smem + buf * CF::STAGEB;
        const unsigned sa = lds0 + (unsigned)buf * CF::STAGEB;
        if constexpr (STG == 2) {
            const unsigned wbase = (unsigned)(tid & ~63) * 16u;
            const unsigned long long bR = uniform64(curR + (unsigned long long)kb0), bC = uniform64(curC + (unsigned long long)kb0);
#pragma unroll
            for (int it = 0; it < CF::ITR; ++it)
                glds_piece(offR[it], bR, __builtin_amdgcn_readfirstlane(sa + wbase + it * (THREADS * 16u)));
#pragma unroll
            for (int it = 0; it < CF::ITC; ++it)
                glds_piece(offC[it], bC, __builtin_amdgcn_readfirstlane(sa + CF::OPB_R + wbase + it * (THREADS * 16u)));
        } else {
            stage_operand<STG, CF::TR, THREADS>((const char*)curR, r0, a.nR, ldRb, kb0, Kb, s, sa, tid);
            stage_operand<STG, CF::TC, THREADS>((const char*)curC, c0, a.nC, ldCb, kb0, Kb, s + CF::OPB_R, sa + CF::OPB_R, tid);
        }
        if (++st_kin == kt_per_seg) {            // wave-uniform
            st_kin = 0;
            ++st_seg;
            curR = st_seg == 1 ? nR1 : nR2;
            curC = st_seg == 1 ? nC1 : nC2;
        }
    };

    // per-lane LDS byte addresses of the fragment reads (stage 0): row base + swizzled 16-byte chunk of each sub-step
    const unsigned laneR = lds0 + (unsigned)(wr * (WM * 32) + l31) * ROWB;
    const unsigned laneC = lds0 + CF::OPB_R + (unsigned)(wc * (WN * 32) + l31) * ROWB;
    unsigned xk[ROWB / 32];
#pragma unroll
    for (int ks = 0; ks < ROWB / 32; ++ks) xk[ks] = (unsigned)(((2 * ks + hh) ^ ((l31 >> 1) & 7)) * 16);

    const bool banded = EPI == EPI_SIM && a.s_gt64 != nullptr && a.count != nullptr;
    TRACE(1);
    if constexpr (STG == 2) {
        // ---- software-pipelined K loop (fast staging) ---------------------------------------------------------------
        // ring of 2 LDS stages, ONE barrier per K-step, placed in front of the LAST sub-step's MFMAs:
        //   sub-steps 0..2 : issue the ds_reads of the next sub-step, counted lgkmcnt, 2*WM*WN... MFMAs on the current one
        //   sub-step 3     : lgkmcnt(0) (own fragments) + vmcnt(0) (stage kt+1 landed) -> s_barrier (every wave has
        //                    finished READING stage kt and every wave's part of stage kt+1 is visible) -> LDS-DMA of
        //                    stage kt+2 into the slot just released -> ds_reads of sub-step 0 of K-step kt+1 -> MFMAs
        // so LDS latency, DMA issue and barrier skew all sit under a group of MFMAs instead of in front of one.
        // The reads are inline asm (hipcc would sink them next to their uses and keep a single fragment set).
        constexpr int NR = WM + WN, NSUB = ROWB / 32;
        // the big tiles (4 waves of 128x128: one wave per SIMD; 8 waves of 128x64): the deep-prefetch K-loop schedule below
        // (the 8-wave 256x256 tile takes the same schedule: 231 VGPRs with the four fragment sets, still two waves per SIMD; C4
        // banded + S 0.595 -> 0.570 ms, count-only 0.457 -> 0.442 ms in back-to-back runs on one box)
        constexpr bool LONE = ((CF::THREADS == 256 && WM * WN >= 16) || (CF::THREADS == 512 && WM * WN == 8)) && MODE != GEMM_F32 && NSUB == 4;
        u32x4 fc[LONE ? 4 : 2][WN], fr[LONE ? 4 : 2][WM];
        auto issue = [&](int kt, int ks, int b) {
#ifdef LAFF_ABL_NOREAD
            if (kt | ks) return;                      // ablation: fragments are read once and reused (wrong results)
#endif
            const unsigned stg_off = (unsigned)(kt & 1) * CF::STAGEB;
            const unsigned ar = laneR + stg_off + xk[ks], ac = laneC + stg_off + xk[ks];
#pragma unroll
            for (int t = 0; t < WN; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fc[b][t]) : "v"(ac), "n"(t * 32 * ROWB));
#pragma unroll
            for (int t = 0; t < WM; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[b][t]) : "v"(ar), "n"(t * 32 * ROWB));
        };
        // LDS-DMA of the stage being refilled, one piece at a time: pieces [0, ITR) are the row operand, the rest the
        // column operand; `fillR/fillC/fill_sa` are the (wave-uniform) source bases and LDS slot of that stage.
        constexpr int NPIECE = CF::ITR + CF::ITC;
        // pieces issued in the barrier sub-step and in the two sub-steps behind it; a lone wave issues the whole refill in two
        constexpr int NP0 = LONE ? (NPIECE + 1) / 2 : (NPIECE + 2) / 3;
        constexpr int NP1 = LONE ? NPIECE - NP0 : (NPIECE - NP0 + 1) / 2;
        constexpr int NP2 = NPIECE - NP0 - NP1;
        unsigned long long fillR = 0, fillC = 0;
        unsigned fill_sa = 0;
        bool fill_on = false;
        const unsigned wbase = (unsigned)(tid & ~63) * 16u;
        auto piece = [&](auto PC) {
            constexpr int pc = decltype(PC)::value;
            if constexpr (pc < CF::ITR)
                glds_piece(offR[pc], fillR, __builtin_amdgcn_readfirstlane(fill_sa + wbase + pc * (THREADS * 16u)));
            else
                glds_piece(offC[pc - CF::ITR], fillC,
                           __builtin_amdgcn_readfirstlane(fill_sa + CF::OPB_R + wbase + (pc - CF::ITR) * (THREADS * 16u)));
        };
        auto mfma_one = [&](int b, auto TRC, auto TCC) {
            constexpr int tr = decltype(TRC)::value, tc = decltype(TCC)::value;
            if constexpr (MODE == GEMM_F32) {
                const f32x4 pa = __builtin_bit_cast(f32x4, fc[b][tc]), pb = __builtin_bit_cast(f32x4, fr[b][tr]);
                acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.x, pb.x, acc[tr][tc], 0, 0, 0);
                acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.y, pb.y, acc[tr][tc], 0, 0, 0);
                acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.z, pb.z, acc[tr][tc], 0, 0, 0);
                acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.w, pb.w, acc[tr][tc], 0, 0, 0);
            } else if constexpr (MODE == GEMM_F16) {
                acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    __builtin_bit_cast(f16x8, fc[b][tc]), __builtin_bit_cast(f16x8, fr[b][tr]), acc[tr][tc], 0, 0, 0);
            } else {
                acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, fc[b][tc]), __builtin_bit_cast(bf16x8, fr[b][tr]), acc[tr][tc], 0, 0, 0);
            }
        };
        // the sub-step's WM*WN MFMA groups with pieces [P0, P0+NP) of the refill slotted in after the first NP groups:
        // a burst of 8 DMA issues right after the barrier kept BOTH waves of a SIMD out of the MFMA pipe for ~600 cycles
        // per K-step (probes: wave 0 waits 1.1k cycles at the barrier, 55 on the DMA landing, 40 on its LDS fragments).
        auto mfmas = [&](int b, auto P0C, auto NPC, bool dma) {
            constexpr int P0 = decltype(P0C)::value, NP = decltype(NPC)::value;
            static_assert(NP <= WM * WN, "");
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (([&] {
                     mfma_one(b, std::integral_constant<int, I / WN>{}, std::integral_constant<int, I % WN>{});
                     if constexpr (I < NP) {
                         __builtin_amdgcn_sched_barrier(0);
                         if (dma) piece(std::integral_constant<int, P0 + I>{});
                         __builtin_amdgcn_sched_barrier(0);
                     }
                 }()),
                 ...);
            }(std::make_integer_sequence<int, WM * WN>{});
        };
        // ---- lone wave per SIMD: nothing else feeds the MFMA pipe while this wave issues anything but an MFMA, so every other
        // instruction of the loop sits in the shadow of one (an MFMA holds the pipe for 32 cycles): the NR fragment reads of a later
        // sub-step go one at a time behind the first NR MFMAs of the current one, the refill's DMA pieces behind the following ones
        // (3 instructions each: glds_piece_s), and what is asked for / refilled is a compile-time property of the K-step (kstep).
        constexpr bool ILV = LONE && NP0 <= WM * WN && NP2 == 0;
        auto issue_one = [&](int kt, int ks, int b, auto IDXC) {
            constexpr int idx = decltype(IDXC)::value;
            const unsigned off = (unsigned)(kt & 1) * CF::STAGEB + xk[ks];
            const unsigned ar = laneR + off, ac = laneC + off;
            auto& fcs = fc;
            auto& frs = fr;
            constexpr int ic = idx < WN ? idx : 0, ir = idx < WN ? 0 : idx - WN;
            if constexpr (idx < WN)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fcs[b][ic]) : "v"(ac), "n"(ic * 32 * ROWB));
            else
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frs[b][ir]) : "v"(ar), "n"(ir * 32 * ROWB));
        };
        unsigned fill_base_s = 0;                       // scalar: LDS byte address of this wave's first piece of the stage being refilled
        auto piece_s = [&](auto PC) {
            constexpr int pc = decltype(PC)::value;
            if constexpr (pc < CF::ITR) glds_piece_s<pc * (THREADS * 16)>(offR[pc], fillR, fill_base_s);
            else glds_piece_s<CF::OPB_R + (pc - CF::ITR) * (THREADS * 16)>(offC[pc - CF::ITR], fillC, fill_base_s);
        };
        auto mfmas_ilv = [&](int b, auto RDC, int nkt_, int nks, int nb, auto P0C, auto NPC, auto DMAC) {
            constexpr int P0 = decltype(P0C)::value, NP = decltype(NPC)::value;
            constexpr bool rd = decltype(RDC)::value, dma = decltype(DMAC)::value;
            static_assert(NR <= WM * WN && NP <= WM * WN, "");
            constexpr int PS = WM * WN - NP;            // pieces ride behind the last NP MFMAs (sharing a slot with a read if they must)
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (([&] {
                     mfma_one(b, std::integral_constant<int, I / WN>{}, std::integral_constant<int, I % WN>{});
                     if constexpr (I < NR && rd) {
                         __builtin_amdgcn_sched_barrier(0);
                         issue_one(nkt_, nks, nb, std::integral_constant<int, I>{});
                         __builtin_amdgcn_sched_barrier(0);
                     }
                     if constexpr (I >= PS && dma) {
                         __builtin_amdgcn_sched_barrier(0);
                         piece_s(std::integral_constant<int, P0 + I - PS>{});
                         __builtin_amdgcn_sched_barrier(0);
                     }
                 }()),
                 ...);
            }(std::make_integer_sequence<int, WM * WN>{});
        };
        using IC0 = std::integral_constant<int, 0>;
        static_assert(NSUB % 2 == 0, "fragment buffer parity must repeat every K-step");
        // prologue: stage 0 landed and visible, stage 1 in flight, fragments of (K-step 0, sub-step 0) in flight
        stage_next(0);
        // row inputs of the banded epilogue: one extra DMA piece for waves 0..3, left in flight by this wait (see band_stage_rows)
        int extra = 0;
        if constexpr (EPI == EPI_SIM) extra = band_stage_rows<CF>(a, r0, c0, wave, lane, lds0);
        if (extra == 1 && nkt > 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");       // wave-uniform
        else if (extra == 2 && nkt > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TRACE(2);
        if (nkt > 1) stage_next(1);
        issue(0, 0, 0);
#ifdef LAFF_GEMM_TRACE
        unsigned long long tw_lds = 0, tw_vm = 0, tw_bar = 0;
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_readcyclecounter();
#endif
        if constexpr (ILV) {
            // Fragments are asked for TWO sub-steps ahead into four register sets (a lone wave has the registers: its accumulators
            // live in the AGPR half), the barrier sits in front of sub-step 2 (every fragment of the K-step has arrived by then, so the
            // stage is released there) and the whole refill is issued in sub-steps 2 and 3: it has 1.5 K-steps to land.
            // Measured at 8192^2 x 4096 (tools/debug/trace_longk.py): 2,538 cycles per K-step against 2,828 for the 8-wave loop
            // (MFMA issue alone 2,048); waits of wave 0 per K-step: fragments 39, DMA landing 46, barrier 44 (8-wave: 39 / 86 / 782).
            using I2 = std::integral_constant<int, NP0>;
            using I3 = std::integral_constant<int, NP1>;
            using T = std::true_type;
            using F = std::false_type;
            issue(0, 1, 1);
            const unsigned wbase_s = __builtin_amdgcn_readfirstlane(wbase);
            // one K-step; MORE: there is a next K-step (its first fragments are asked for here), FILL: and one after that (refilled here)
            auto kstep = [&](int kt, auto MOREC, auto FILLC) {
                constexpr bool more = decltype(MOREC)::value, fill = decltype(FILLC)::value;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                mfmas_ilv(0, T{}, kt, 2, 2, IC0{}, IC0{}, F{});
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                mfmas_ilv(1, T{}, kt, 3, 3, IC0{}, IC0{}, F{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (more) {
#ifdef LAFF_GEMM_TRACE
                    const unsigned long long t_a = __builtin_readcyclecounter();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const unsigned long long t_b = __builtin_readcyclecounter();
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    const unsigned long long t_c = __builtin_readcyclecounter();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    const unsigned long long t_d = __builtin_readcyclecounter();
                    tw_lds += t_b - t_a; tw_vm += t_c - t_b; tw_bar += t_d - t_c;
#else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#endif
                    if constexpr (fill) {                                // slot of stage kt is free now: refill it
                        fill_base_s = lds0 + (unsigned)(kt & 1) * CF::STAGEB + wbase_s;
                        const unsigned long long kb0 = (unsigned long long)((long)st_kin * ROWB);
                        fillR = uniform64(curR + kb0);
                        fillC = uniform64(curC + kb0);
                        if (++st_kin == kt_per_seg) {                    // wave-uniform
                            st_kin = 0;
                            ++st_seg;
                            curR = st_seg == 1 ? nR1 : nR2;
                            curC = st_seg == 1 ? nC1 : nC2;
                        }
                    }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                mfmas_ilv(2, MOREC, kt + 1, 0, 0, IC0{}, I2{}, FILLC);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (more) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                mfmas_ilv(3, MOREC, kt + 1, 1, 1, I2{}, I3{}, FILLC);
                __builtin_amdgcn_sched_barrier(0);
            };
            const unsigned m0_keep = m0_save();
            int kt = 0;
            for (; kt + 2 < nkt; ++kt) {
                if (kt == 1) TRACE(3);
                kstep(kt, T{}, T{});
            }
            if (kt + 1 < nkt) { kstep(kt, T{}, F{}); ++kt; }
            kstep(kt, F{}, F{});
            m0_restore(m0_keep);
        } else {
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt == 1) TRACE(3);
            // sub-steps 0 .. NSUB-2; the first two also carry the rest of the refill started at the previous barrier
#pragma unroll
            for (int ks = 0; ks < NSUB - 1; ++ks) {
                issue(kt, ks + 1, (ks + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0) mfmas(ks & 1, std::integral_constant<int, NP0>{}, std::integral_constant<int, NP1>{}, fill_on);
                else if (ks == 1) mfmas(ks & 1, std::integral_constant<int, NP0 + NP1>{}, std::integral_constant<int, NP2>{}, fill_on);
                else mfmas(ks & 1, IC0{}, IC0{}, false);
                __builtin_amdgcn_sched_barrier(0);
            }
            // last sub-step: its fragments are the only LDS reads outstanding
            fill_on = false;
            if (kt + 1 < nkt) {
#ifdef LAFF_GEMM_TRACE
                // wait breakdown of wave 0 (debug build): LDS fragments / LDS-DMA landing / barrier skew, summed over the K loop
                const unsigned long long t_a = __builtin_readcyclecounter();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long t_b = __builtin_readcyclecounter();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const unsigned long long t_c = __builtin_readcyclecounter();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const unsigned long long t_d = __builtin_readcyclecounter();
                tw_lds += t_b - t_a; tw_vm += t_c - t_b; tw_bar += t_d - t_c;
#else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#endif
                if (kt + 2 < nkt) {                                  // slot of stage kt is free now: start refilling it
#ifndef LAFF_ABL_NODMA
                    fill_on = true;
#endif
                    fill_sa = lds0 + (unsigned)(kt & 1) * CF::STAGEB;
                    const unsigned long long kb0 = (unsigned long long)((long)st_kin * ROWB);
                    fillR = uniform64(curR + kb0);
                    fillC = uniform64(curC + kb0);
                    if (++st_kin == kt_per_seg) {                    // wave-uniform
                        st_kin = 0;
                        ++st_seg;
                        curR = st_seg == 1 ? nR1 : nR2;
                        curC = st_seg == 1 ? nC1 : nC2;
                    }
                }
                issue(kt + 1, 0, 0);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas((NSUB - 1) & 1, IC0{}, std::integral_constant<int, NP0>{}, fill_on);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
#ifdef LAFF_GEMM_TRACE
        if (a.trace && tid == 0) {
            unsigned long long* t2 = a.trace + (long)gridDim.x * 8 + (long)blockIdx.x * 8;
            t2[0] = tw_lds; t2[1] = tw_vm; t2[2] = tw_bar; t2[3] = (unsigned long long)nkt;
            t2[4] = rt0; t2[5] = __builtin_amdgcn_s_memrealtime(); t2[6] = ct0; t2[7] = __builtin_readcyclecounter();
        }
#endif
    } else {
        // ---- generic staging paths: compiler-scheduled loop, one barrier pair per K-step ------------------------------
        if constexpr (EPI == EPI_SIM) (void)band_stage_rows<CF>(a, r0, c0, wave, lane, lds0);   // landed + visible after the first barrier
        stage_next(0);
        for (int kt = 0; kt < nkt; ++kt) {
            if constexpr (GLDS) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            } else {
                __syncthreads();
            }
            if (kt + 1 < nkt) stage_next((kt + 1) & 1);
            const char* sR = smem + (kt & 1) * CF::STAGEB;
            const char* sC = sR + CF::OPB_R;
#pragma unroll
            for (int ks = 0; ks < ROWB / 32; ++ks) {
                const int chunk = 2 * ks + hh;
                uint4 fc[WN], fr[WM];
#pragma unroll
                for (int t = 0; t < WN; ++t) fc[t] = lds_frag(sC, wc * (WN * 32) + t * 32 + l31, chunk);
#pragma unroll
                for (int t = 0; t < WM; ++t) fr[t] = lds_frag(sR, wr * (WM * 32) + t * 32 + l31, chunk);
#pragma unroll
                for (int tr = 0; tr < WM; ++tr)
#pragma unroll
                    for (int tc = 0; tc < WN; ++tc) {
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
            if constexpr (!GLDS) __syncthreads();
        }
    }

    TRACE(4);
    __syncthreads();                                   // every wave is done reading the operand ring
    TRACE(5);
    const bool full = (r0 + CF::TR <= a.nR) && (c0 + CF::TC <= a.nC) && ((a.ldo & 3) == 0) &&
                      ((((uintptr_t)a.out) & 15) == 0);
    if (EPI == EPI_SIM && banded) {
        if (full) epilogue_banded<true, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
        else epilogue_banded<false, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
    } else if constexpr (EPI == EPI_FC) {
        if (full) epilogue_fc<true, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
        else epilogue_fc<false, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
    } else if (full) epilogue<EPI, true, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
    else epilogue<EPI, false, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
    TRACE(6);
#undef TRACE
}

// ---- hi/lo split products with the three partial products INTERLEAVED per K-step ("x3" tile) -----------------------
// The virtual-K concatenation above streams [lo|hi|hi] . [hi|lo|hi]: every hi plane crosses L2 -> LDS -> registers twice.
// Here a stage holds the four planes (R_hi, R_lo, C_hi, C_lo) of ONE 32-element K-step (64-byte rows, 4 x 16 KiB) and each
// 16-element slice runs three MFMA groups on them:
//     B: R_hi . C_lo        C: R_hi . C_hi        A: R_lo . C_hi
// with fragment sets arranged so that C reuses B's R_hi fragments and A reuses C's C_hi fragments: 12 fragment reads per
// 24 MFMAs (0.5 per MFMA instead of 0.75), 64 KiB of LDS-DMA per 48 MFMAs per wave instead of per 32, one barrier per 48.
// LDS reads and the refill DMA are what bound the K loop, so this is where the split GEMM gains.
struct CfgX3 {
    static constexpr int WM = 4, WN = 2, WR = 2, WC = 4;
    static constexpr int TR = 256, TC = 256, THREADS = 512;
    static constexpr int ROWB = 64, CPR = 4;                       // one K-step = 32 16-bit elements
    static constexpr int PLB = TR * ROWB;                          // one plane of one operand: 16 KiB
    static constexpr int OPB_R = 2 * PLB, OPB_C = 2 * PLB, STAGEB = OPB_R + OPB_C, SMEM = 2 * STAGEB;
    static constexpr int PITCH = WN * 32 + 4;
    static constexpr int WPS = 2;
    static_assert(THREADS / 64 * 32 * PITCH * 4 <= SMEM, "epilogue slabs must fit in the operand ring");
};

// RF32: the row operand is given as fp32 (a.Rf, leading dimension a.ldRf, per-row power-of-two scales a.row_scale from
// laff_row_scales) and split into its fp16 hi / lo planes HERE, on its way into LDS -- the planes are never written to HBM
// (laff_split_rows writes 410 MB and the GEMM reads them back at C4).  16 values per thread and K-step: two 32-byte global
// loads right after the barrier, converted (same arithmetic as split_rows_kernel: bit-identical planes) and stored with four
// ds_write_b128 one K-step later, between two MFMA groups.
template <int MODE, int EPI, bool RF32 = false>
__device__ __forceinline__ void gemm_tile_x3(const GemmArgs& a, const int r0, const int c0, char* smem) {
    static_assert(MODE == GEMM_F16 || MODE == GEMM_BF16, "the split product runs on the 16-bit matrix pipe");
    using CF = CfgX3;
    constexpr int WM = CF::WM, WN = CF::WN, THREADS = CF::THREADS, RB = CF::ROWB, CPR = CF::CPR;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / CF::WC, wc = wave % CF::WC;
    const int l31 = lane & 31, hh = lane >> 5;

    const long ldRb = (long)a.ldR * 2, ldCb = (long)a.ldC * 2;
    const int nkt = (int)(((long)a.K * 2) / RB);                  // K bytes are a multiple of 64 on this path (128 unless RF32)
#ifdef LAFF_GEMM_TRACE
#define XTRACE(i) do { if (a.trace && tid == 0) a.trace[(long)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    if (a.trace && tid == 0)
        a.trace[(long)blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xf814) /*XCC_ID*/ << 32) |
                                           __builtin_amdgcn_s_getreg(0xf804) /*HW_ID*/;
#else
#define XTRACE(i) do {} while (0)
#endif
    XTRACE(0);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    // per-lane source offsets inside a plane: slot p = sub*THREADS + tid -> (row p/4, swizzled chunk); the same for hi and lo
    unsigned offR[2], offC[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int p = sub * THREADS + tid, row = p / CPR, ch = (p % CPR) ^ ((row >> 2) & 3);
        offR[sub] = (unsigned)min(r0 + row, a.nR - 1) * (unsigned)ldRb + (unsigned)ch * 16u;
        offC[sub] = (unsigned)min(c0 + row, a.nC - 1) * (unsigned)ldCb + (unsigned)ch * 16u;
    }
    // RF32: unit u = sub*THREADS + tid covers 8 consecutive K values (32 source bytes) of row u/4
    unsigned xoff[2] = {0, 0}, xdst[2] = {0, 0};
    float xscale[2] = {1.0f, 1.0f};
    u32x4 xs[2][2];
    unsigned long long fX = 0;
    unsigned x_sa = 0;                                            // LDS slot the staged values go to
    if constexpr (RF32) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int p = sub * THREADS + tid, row = p / CPR, ch = p % CPR;
            const int gr = min(r0 + row, a.nR - 1);
            xoff[sub] = (unsigned)gr * (unsigned)(a.ldRf * 4) + (unsigned)ch * 32u;
            xdst[sub] = (unsigned)row * RB + (unsigned)(ch ^ ((row >> 2) & 3)) * 16u;
            xscale[sub] = 1.0f / a.row_scale[gr];                 // exact: the scales are powers of two
        }
    }
    // planes: segments of the concatenated formulation are (R_lo, C_hi), (R_hi, C_lo), (R_hi, C_hi)
    const unsigned long long bRhi = (unsigned long long)((const char*)a.R + a.segR[1]), bRlo = (unsigned long long)((const char*)a.R + a.segR[0]);
    const unsigned long long bChi = (unsigned long long)((const char*)a.C + a.segC[0]), bClo = (unsigned long long)((const char*)a.C + a.segC[1]);
    unsigned long long fRhi = 0, fRlo = 0, fChi = 0, fClo = 0;     // wave-uniform source bases of the stage being filled
    unsigned fill_sa = 0, fill_base_s = 0;
    const unsigned wbase = (unsigned)(tid & ~63) * 16u;
    auto fill_begin = [&](int kt, int buf) {
        const unsigned long long kb = (unsigned long long)((long)kt * RB);
        fRhi = uniform64(bRhi + kb); fRlo = uniform64(bRlo + kb); fChi = uniform64(bChi + kb); fClo = uniform64(bClo + kb);
        fill_sa = lds0 + (unsigned)buf * CF::STAGEB;
        fill_base_s = __builtin_amdgcn_readfirstlane(fill_sa + wbase);
        if constexpr (RF32) fX = uniform64((unsigned long long)((const char*)a.Rf) + (unsigned long long)((long)kt * (RB * 2)));
    };
    auto xload = [&]() {                         // 4 x global_load_dwordx4 (saddr form), no wait
        x_sa = fill_sa;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16"
                         : "=&v"(xs[sub][0]), "=&v"(xs[sub][1])
                         : "v"(xoff[sub]), "s"(fX)
                         : "memory");
        }
    };
    auto xconvert = [&]() {                      // xs -> fp16 hi / lo chunks in the slot recorded by xload
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const f32x4 lo4 = __builtin_bit_cast(f32x4, xs[sub][0]), hi4 = __builtin_bit_cast(f32x4, xs[sub][1]);
            const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            h8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = v[e] * xscale[sub];
                h[e] = (_Float16)t;
                l[e] = (_Float16)(t - (float)h[e]);
            }
            const unsigned d = x_sa + xdst[sub];
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:%3"
                         :: "v"(d), "v"(__builtin_bit_cast(u32x4, h)), "v"(__builtin_bit_cast(u32x4, l)), "n"(CF::PLB)
                         : "memory");
        }
    };
    // piece pc = operand*4 + plane*2 + sub (operand 0 = R, plane 0 = hi)
    auto piece = [&](auto PC) {
        constexpr int pc0 = decltype(PC)::value;
        if constexpr (RF32 && pc0 >= 4) return;                   // only the four column-operand pieces exist; they go first
        constexpr int pc = RF32 ? pc0 + 4 : pc0, op = pc >> 2, plane = (pc >> 1) & 1, sub = pc & 1;
        const unsigned dst = fill_sa + (unsigned)(op * CF::OPB_R + plane * CF::PLB + sub * (THREADS * 16)) + wbase;
        const unsigned long long base = op == 0 ? (plane == 0 ? fRhi : fRlo) : (plane == 0 ? fChi : fClo);
        glds_piece(op == 0 ? offR[sub] : offC[sub], base, __builtin_amdgcn_readfirstlane(dst));
    };
    // the same piece with the LDS address as scalar base + constant and M0 left pointing at it (3 instructions instead of 6: in the K
    // loop every issue slot that is not an MFMA is a bubble the other wave of the SIMD has to fill)
    auto piece_s = [&](auto PC) {
        constexpr int pc0 = decltype(PC)::value;
        if constexpr (RF32 && pc0 >= 4) return;
        constexpr int pc = RF32 ? pc0 + 4 : pc0, op = pc >> 2, plane = (pc >> 1) & 1, sub = pc & 1;
        const unsigned long long base = op == 0 ? (plane == 0 ? fRhi : fRlo) : (plane == 0 ? fChi : fClo);
        glds_piece_s<op * CF::OPB_R + plane * CF::PLB + sub * (THREADS * 16)>(op == 0 ? offR[sub] : offC[sub], base, fill_base_s);
    };
    auto fill_all = [&]() {
        piece(std::integral_constant<int, 0>{}); piece(std::integral_constant<int, 1>{});
        piece(std::integral_constant<int, 2>{}); piece(std::integral_constant<int, 3>{});
        piece(std::integral_constant<int, 4>{}); piece(std::integral_constant<int, 5>{});
        piece(std::integral_constant<int, 6>{}); piece(std::integral_constant<int, 7>{});
    };

    // fragment reads: row base + swizzled chunk (2*ks + hh) of the 64-byte row; plane / stage / 32-row block as offsets
    const unsigned laneR = lds0 + (unsigned)(wr * (WM * 32) + l31) * RB;
    const unsigned laneC = lds0 + CF::OPB_R + (unsigned)(wc * (WN * 32) + l31) * RB;
    unsigned xk[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xk[ks] = (unsigned)(((2 * ks + hh) ^ ((l31 >> 2) & 3)) * 16);
    u32x4 fc[2][WN], fr[2][WM];
    auto rd_fc = [&](int set, int kt, int ks, int plane) {
        const unsigned ac = laneC + (unsigned)(kt & 1) * CF::STAGEB + (unsigned)plane * CF::PLB + xk[ks];
#pragma unroll
        for (int t = 0; t < WN; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fc[set][t]) : "v"(ac), "n"(t * 32 * RB));
    };
    auto rd_fr = [&](int set, int kt, int ks, int plane) {
        const unsigned ar = laneR + (unsigned)(kt & 1) * CF::STAGEB + (unsigned)plane * CF::PLB + xk[ks];
#pragma unroll
        for (int t = 0; t < WM; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[set][t]) : "v"(ar), "n"(t * 32 * RB));
    };
    // WM*WN MFMAs on (fc[FS], fr[RS]) with refill pieces [P0, P0+NP) slotted in after the first NP of them
    auto mm = [&](auto FSC, auto RSC, auto P0C, auto NPC, auto DMAC) {
        constexpr bool dma = decltype(DMAC)::value;
        constexpr int FS = decltype(FSC)::value, RS = decltype(RSC)::value, P0 = decltype(P0C)::value, NP = decltype(NPC)::value;
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (([&] {
                 constexpr int tr = I / WN, tc = I % WN;
                 if constexpr (MODE == GEMM_F16)
                     acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fc[FS][tc]),
                                                                          __builtin_bit_cast(f16x8, fr[RS][tr]), acc[tr][tc], 0, 0, 0);
                 else
                     acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fc[FS][tc]),
                                                                           __builtin_bit_cast(bf16x8, fr[RS][tr]), acc[tr][tc], 0, 0, 0);
                 if constexpr (I < NP && dma) {
                     __builtin_amdgcn_sched_barrier(0);
                     piece_s(std::integral_constant<int, P0 + I>{});
                     __builtin_amdgcn_sched_barrier(0);
                 }
             }()),
             ...);
        }(std::make_integer_sequence<int, WM * WN>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I6 = std::integral_constant<int, 6>;

    const bool banded = EPI == EPI_SIM && a.s_gt64 != nullptr && a.count != nullptr;
    XTRACE(1);
    // prologue: stage 0 landed and visible, stage 1 in flight, operands of the first B group (C_lo, R_hi of slice 0) in flight
    fill_begin(0, 0);
    if constexpr (RF32) xload();
    fill_all();
    int extra = 0;
    if constexpr (EPI == EPI_SIM) extra = band_stage_rows<CF>(a, r0, c0, wave, lane, lds0);     // see band_stage_rows
    if constexpr (RF32) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[1][0]), "+v"(xs[1][1])::"memory");
        xconvert();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
        if (extra && nkt > 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");     // wave-uniform
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nkt > 1) {
        fill_begin(1, 1);
        if constexpr (RF32) xload();
        fill_all();
    }
    rd_fc(1, 0, 0, 1);
    rd_fr(1, 0, 0, 0);
    XTRACE(2);
    // one K-step with its place in the loop as compile-time properties -- MORE: there is a next K-step (barrier, first fetches from its
    // stage); FILL: and one after that (its refill starts at this K-step's barrier: pieces 0..3 here, 4..7 in the next K-step, which
    // therefore has PREV); with RF32, MORE also means "the X rows of the next K-step are waiting to be converted".  The peeled head /
    // tail leave no per-slot scalar branch in the steady-state K-step.
    using T = std::true_type;
    using F = std::false_type;
    auto kstep = [&](int kt, auto MOREC, auto FILLC, auto PREVC) {
        constexpr bool more = decltype(MOREC)::value, fill = decltype(FILLC)::value;
        // ======== slice 0
        // ---- B: R_hi . C_lo (set 1); meanwhile fetch C_hi -> fc[0], R_lo -> fr[0] of this slice
        rd_fc(0, kt, 0, 0);
        rd_fr(0, kt, 0, 1);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        mm(I1{}, I1{}, I4{}, I2{}, PREVC);                         // pieces 4,5 of the refill started at the previous barrier
        // ---- C: R_hi . C_hi (fr[1], fc[0]); meanwhile fetch slice 1's C_lo -> fc[1]
        rd_fc(1, kt, 1, 1);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        mm(I0{}, I1{}, I6{}, I2{}, PREVC);                         // pieces 6,7
        // ---- A: R_lo . C_hi (fr[0], fc[0]); meanwhile fetch slice 1's R_hi -> fr[1]
        rd_fr(1, kt, 1, 0);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        mm(I0{}, I0{}, I0{}, I0{}, F{});
        if constexpr (RF32 && more) {
            // the X loads are older than the four column-operand pieces issued since: vmcnt(4) = "X has landed"
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[1][0]), "+v"(xs[1][1])::"memory");
            xconvert();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ======== slice 1
        rd_fc(0, kt, 1, 0);
        rd_fr(0, kt, 1, 1);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        mm(I1{}, I1{}, I0{}, I0{}, F{});
        if constexpr (more) {
            // every wave has read all it needs from stage kt (the fetches above have landed) and stage kt+1 has landed
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (fill) {
                fill_begin(kt + 2, kt & 1);
                if constexpr (RF32) xload();
            }
            rd_fc(1, kt + 1, 0, 1);
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        }
        mm(I0{}, I1{}, I0{}, I2{}, FILLC);                         // pieces 0,1 of the refill just started
        if constexpr (more) {
            rd_fr(1, kt + 1, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        mm(I0{}, I0{}, I2{}, I2{}, FILLC);                         // pieces 2,3
    };
    {
        const unsigned m0_keep = m0_save();
        int kt = 0;
        if (nkt >= 3) {
            kstep(0, T{}, T{}, F{});
            for (kt = 1; kt + 2 < nkt; ++kt) {
                if (kt == 1) XTRACE(3);
                kstep(kt, T{}, T{}, T{});
            }
            kstep(kt, T{}, F{}, T{});
            ++kt;
        } else if (nkt == 2) {
            kstep(0, T{}, F{}, F{});
            kt = 1;
        }
        kstep(kt, F{}, F{}, F{});
        m0_restore(m0_keep);
    }

    XTRACE(4);
    __syncthreads();                                   // every wave is done reading the operand ring
    XTRACE(5);
    const bool full = (r0 + CF::TR <= a.nR) && (c0 + CF::TC <= a.nC) && ((a.ldo & 3) == 0) &&
                      ((((uintptr_t)a.out) & 15) == 0);
    if (EPI == EPI_SIM && banded) {
        if (full) epilogue_banded<true, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
        else epilogue_banded<false, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
    } else if constexpr (EPI == EPI_FC) {
        if (full) epilogue_fc<true, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
        else epilogue_fc<false, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
    } else if (full) epilogue<EPI, true, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
    else epilogue<EPI, false, CF>(a, acc, r0, c0, wr, wc, wave, lane, smem);
    XTRACE(6);
#undef XTRACE
}

template <int MODE>
__global__ __launch_bounds__(CfgX3::THREADS, CfgX3::WPS) void gemm_nt_x3_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_r = (a.nR + CfgX3::TR - 1) / CfgX3::TR, tiles_c = (a.nC + CfgX3::TC - 1) / CfgX3::TC;
    int r0, c0;
    tile_origin<CfgX3>(a, xcd_remap(blockIdx.x, tiles_r * tiles_c), r0, c0);
    gemm_tile_x3<MODE, EPI_SIM>(a, r0, c0, smem);
}

template <int MODE>
__global__ __launch_bounds__(CfgX3::THREADS, CfgX3::WPS) void gemm_nt_x3_fused_grouped_kernel(GroupedGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lin = xcd_remap(blockIdx.x, g.tile_start[g.count]);
    int p = 0;
    while (p + 1 < g.count && lin >= g.tile_start[p + 1]) ++p;     // wave-uniform scalar search
    int r0, c0;
    tile_origin<CfgX3>(g.p[p], lin - g.tile_start[p], r0, c0);
    gemm_tile_x3<MODE, EPI_FC, true>(g.p[p], r0, c0, smem);
}

template <int MODE>
__global__ __launch_bounds__(CfgX3::THREADS, CfgX3::WPS) void gemm_nt_x3_grouped_kernel(GroupedGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Blocks [0, nbig) run 256x256 tiles.  The big tiles that would form a sparsely filled LAST round (e.g. 40 of 256 CUs busy
    // for a whole tile time: 12 % of this launch at C4) are cut into four 128x128 tiles each, run by blocks [nbig, grid) on the
    // Cfg128 body (waves 4..7 of those blocks retire at once): the tail round then lasts a quarter as long.
    const bool big = (int)blockIdx.x < g.nbig;
    const int lin = big ? xcd_remap(blockIdx.x, g.nbig) : g.nbig + (((int)blockIdx.x - g.nbig) >> 2);
    int p = 0;
    while (p + 1 < g.count && lin >= g.tile_start[p + 1]) ++p;     // wave-uniform scalar search
    int r0, c0;
    tile_origin<CfgX3>(g.p[p], lin - g.tile_start[p], r0, c0);
    if (big) {
        gemm_tile_x3<MODE, EPI_FC>(g.p[p], r0, c0, smem);
    } else {
        const int q = ((int)blockIdx.x - g.nbig) & 3;
        r0 += (q >> 1) * Cfg128::TR;
        c0 += (q & 1) * Cfg128::TC;
        if (r0 >= g.p[p].nR || c0 >= g.p[p].nC || threadIdx.x >= Cfg128::THREADS) return;
        gemm_tile<MODE, 2, Cfg128, EPI_FC>(g.p[p], r0, c0, smem);
    }
}

template <int MODE, int STG, typename CF>
__global__ __launch_bounds__(CF::THREADS, CF::WPS) void gemm_nt_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_r = (a.nR + CF::TR - 1) / CF::TR, tiles_c = (a.nC + CF::TC - 1) / CF::TC;
    int r0, c0;
    tile_origin<CF>(a, xcd_remap(blockIdx.x, tiles_r * tiles_c), r0, c0);
    gemm_tile<MODE, STG, CF, EPI_SIM>(a, r0, c0, smem);
}

// several independent problems (the FC projections of all fused features) in ONE launch: fills the chip where a
// single 10k-row projection has only 316 tiles for 512 workgroup slots, and removes 7 launch boundaries.
template <int MODE, int STG, typename CF>
__global__ __launch_bounds__(CF::THREADS, CF::WPS) void gemm_nt_grouped_kernel(GroupedGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lin = xcd_remap(blockIdx.x, g.tile_start[g.count]);
    int p = 0;
    while (p + 1 < g.count && lin >= g.tile_start[p + 1]) ++p;     // wave-uniform scalar search
    int r0, c0;
    tile_origin<CF>(g.p[p], lin - g.tile_start[p], r0, c0);
    gemm_tile<MODE, STG, CF, EPI_FC>(g.p[p], r0, c0, smem);
}

template <int MODE, int STG, typename CF>
static hipError_t launch_t(const GemmArgs& a, hipStream_t st) {
    const long nb = (long)((a.nR + CF::TR - 1) / CF::TR) * ((a.nC + CF::TC - 1) / CF::TC);
    if (nb <= 0 || nb > 0x7fffffffL) return hipErrorInvalidValue;
    static unsigned long long attr_done = 0;     // per device; also keeps the call out of HIP-graph captures
    if (hipError_t e = smem_attr_once(attr_done, gemm_nt_kernel<MODE, STG, CF>, CF::SMEM + PAIR_LDS); e != hipSuccess) return e;
    hipLaunchKernelGGL((gemm_nt_kernel<MODE, STG, CF>), dim3((unsigned)nb), dim3(CF::THREADS), CF::SMEM + PAIR_LDS, st, a);
    return hipGetLastError();
}

// staging kind of one problem: 0 = through registers (unaligned rows / ragged K), 1 = LDS-DMA with per-step address
// arithmetic (K tail in whole 16-byte chunks), 2 = LDS-DMA fast path (K bytes a multiple of the K-step, operands < 4 GiB)
int staging_kind(const GemmArgs& a, int esz, bool aligned) {
    if (!aligned) return 0;
    const long long Kb = (long long)a.K * esz;
    const long long spanR = (long long)a.nR * a.ldR * esz, spanC = (long long)a.nC * a.ldC * esz;
    if (Kb % ROWB == 0 && spanR < (1ll << 32) && spanC < (1ll << 32)) return 2;
    return 1;
}

template <int MODE>
static hipError_t launch_x3(const GemmArgs& a, hipStream_t st);

int g_num_cus = 256;       // set from the device properties when a ctx is created
int g_gemm_variant = 0;   // tuning knob LAFF_GEMM_VARIANT: 128 / 256 / 512 force the tile configuration of the 16-bit GEMM (256 also
                          // keeps split products in the concatenated form); 3 forces the big tiles incl. the interleaved x3 tile

template <int MODE>
static hipError_t launch_m(const GemmArgs& a, bool aligned, hipStream_t st) {
    const int stg = staging_kind(a, ModeTraits<MODE>::ESZ, aligned);
    if (stg == 0) return launch_t<MODE, 0, Cfg128>(a, st);
    if (stg == 1) return launch_t<MODE, 1, Cfg128>(a, st);
    if constexpr (MODE != GEMM_F32) {
        // big tiles when there are enough of them to fill 256 CUs a few times over
        const long tiles256 = (long)((a.nR + 255) / 256) * ((a.nC + 255) / 256);
        const bool big = g_gemm_variant == 256 || g_gemm_variant == 3 || (g_gemm_variant != 128 && tiles256 >= 512);
        if (big && a.nseg == 3 && g_gemm_variant != 256) return launch_x3<MODE>(a, st);
        // long K, many tiles: 4 waves of 128x128 -- 1/3 fewer LDS fragment reads per MFMA and a K loop scheduled for a lone wave per
        // SIMD: 2,538 against 2,828 cycles per K-step at K = 4096 (tools/debug/trace_longk.py).  Its prologue and epilogue are longer
        // (4 waves do the work of 8; the banded epilogue reads its accumulators out of the AGPR half), so it pays where the K loop
        // dominates (tools/debug/time_shape.py, banded + S): 100k x 30k x 4096 bf16 23.5 -> 22.3 ms, count-only 21.7 -> 20.0 ms;
        // 59,800 x 2,990 x 4096 1.54 -> 1.59 ms (not taken: 2,808 tiles); 16384^2 x 2048 1.06 -> 1.07 ms (not taken).
        const bool lone = g_gemm_variant == 512 || (g_gemm_variant == 0 && a.nseg == 1 && tiles256 >= 4096 &&
                                                    (long long)a.K * ModeTraits<MODE>::ESZ >= 8192);
        if (lone) return launch_t<MODE, 2, Cfg256L>(a, st);
        if (big) return launch_t<MODE, 2, Cfg256>(a, st);
    }
    return launch_t<MODE, 2, Cfg128>(a, st);
}

template <int STG>
static hipError_t launch_grouped_t(GroupedGemmArgs& g, hipStream_t st) {
    long nb = 0;
    for (int i = 0; i < g.count; ++i) {
        g.tile_start[i] = (int)nb;
        nb += (long)((g.p[i].nR + Cfg128::TR - 1) / Cfg128::TR) * ((g.p[i].nC + Cfg128::TC - 1) / Cfg128::TC);
    }
    if (nb <= 0 || nb > 0x7fffffffL) return hipErrorInvalidValue;
    g.tile_start[g.count] = (int)nb;
    hipLaunchKernelGGL((gemm_nt_grouped_kernel<GEMM_F32, STG, Cfg128>), dim3((unsigned)nb), dim3(Cfg128::THREADS), Cfg128::SMEM, st, g);
    return hipGetLastError();
}

template <typename CF>
static hipError_t launch_grouped_f16_t(GroupedGemmArgs& g, hipStream_t st) {
    long nb = 0;
    for (int i = 0; i < g.count; ++i) {
        g.tile_start[i] = (int)nb;
        nb += (long)((g.p[i].nR + CF::TR - 1) / CF::TR) * ((g.p[i].nC + CF::TC - 1) / CF::TC);
    }
    if (nb <= 0 || nb > 0x7fffffffL) return hipErrorInvalidValue;
    g.tile_start[g.count] = (int)nb;
    static unsigned long long attr_done = 0;     // per device; also keeps the call out of HIP-graph captures
    if (hipError_t e = smem_attr_once(attr_done, gemm_nt_grouped_kernel<GEMM_F16, 2, CF>, CF::SMEM); e != hipSuccess) return e;
    hipLaunchKernelGGL((gemm_nt_grouped_kernel<GEMM_F16, 2, CF>), dim3((unsigned)nb), dim3(CF::THREADS), CF::SMEM, st, g);
    return hipGetLastError();
}

static hipError_t launch_grouped_x3(GroupedGemmArgs& g, hipStream_t st) {
    long nb = 0;
    for (int i = 0; i < g.count; ++i) {
        g.tile_start[i] = (int)nb;
        nb += (long)((g.p[i].nR + CfgX3::TR - 1) / CfgX3::TR) * ((g.p[i].nC + CfgX3::TC - 1) / CfgX3::TC);
    }
    if (nb <= 0 || nb > 0x7fffffffL) return hipErrorInvalidValue;
    g.tile_start[g.count] = (int)nb;
    // tail split: the big tiles of a last round that would fill at most 3/4 of the CUs become 4 small tiles each
    const long rem = nb % g_num_cus;
    g.nbig = (int)nb;
    if (g_gemm_variant != 4 && nb > g_num_cus && rem > 0 && rem * 4 <= 3L * g_num_cus) g.nbig = (int)(nb - rem);
    const long grid = g.nbig + 4L * (nb - g.nbig);
    static unsigned long long attr_done = 0;     // per device; also keeps the call out of HIP-graph captures
    if (hipError_t e = smem_attr_once(attr_done, gemm_nt_x3_grouped_kernel<GEMM_F16>, CfgX3::SMEM); e != hipSuccess) return e;
    hipLaunchKernelGGL((gemm_nt_x3_grouped_kernel<GEMM_F16>), dim3((unsigned)grid), dim3(CfgX3::THREADS), CfgX3::SMEM, st, g);
    return hipGetLastError();
}

hipError_t launch_gemm_nt_x3_fused_grouped(GroupedGemmArgs& g, hipStream_t st) {
    long nb = 0;
    for (int i = 0; i < g.count; ++i) {
        g.tile_start[i] = (int)nb;
        nb += (long)((g.p[i].nR + CfgX3::TR - 1) / CfgX3::TR) * ((g.p[i].nC + CfgX3::TC - 1) / CfgX3::TC);
    }
    if (nb <= 0 || nb > 0x7fffffffL) return hipErrorInvalidValue;
    g.tile_start[g.count] = (int)nb;
    g.nbig = (int)nb;
    static unsigned long long attr_done = 0;     // per device; also keeps the call out of HIP-graph captures
    if (hipError_t e = smem_attr_once(attr_done, gemm_nt_x3_fused_grouped_kernel<GEMM_F16>, CfgX3::SMEM); e != hipSuccess) return e;
    hipLaunchKernelGGL((gemm_nt_x3_fused_grouped_kernel<GEMM_F16>), dim3((unsigned)nb), dim3(CfgX3::THREADS), CfgX3::SMEM, st, g);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_x3(const GemmArgs& a, hipStream_t st) {
    const long nb = (long)((a.nR + CfgX3::TR - 1) / CfgX3::TR) * ((a.nC + CfgX3::TC - 1) / CfgX3::TC);
    if (nb <= 0 || nb > 0x7fffffffL) return hipErrorInvalidValue;
    static unsigned long long attr_done = 0;     // per device; also keeps the call out of HIP-graph captures
    if (hipError_t e = smem_attr_once(attr_done, gemm_nt_x3_kernel<MODE>, CfgX3::SMEM + PAIR_LDS); e != hipSuccess) return e;
    hipLaunchKernelGGL((gemm_nt_x3_kernel<MODE>), dim3((unsigned)nb), dim3(CfgX3::THREADS), CfgX3::SMEM + PAIR_LDS, st, a);
    return hipGetLastError();
}

hipError_t launch_gemm_nt_grouped_f16(GroupedGemmArgs& g, hipStream_t st) {
    long t256 = 0;
    bool split = true;
    for (int i = 0; i < g.count; ++i) {
        t256 += (long)((g.p[i].nR + 255) / 256) * ((g.p[i].nC + 255) / 256);
        split = split && g.p[i].nseg == 3;
    }
    const bool big = g_gemm_variant == 256 || g_gemm_variant == 3 || (g_gemm_variant != 128 && t256 >= 512);
    if (big && split && g_gemm_variant != 256) return launch_grouped_x3(g, st);     // LAFF_GEMM_VARIANT=256: concatenated form
    return big ? launch_grouped_f16_t<Cfg256>(g, st) : launch_grouped_f16_t<Cfg128>(g, st);
}

hipError_t launch_gemm_nt_grouped_f32(GroupedGemmArgs& g, int stg, hipStream_t st) {
    switch (stg) {
        case 0: return launch_grouped_t<0>(g, st);
        case 1: return launch_grouped_t<1>(g, st);
        default: return launch_grouped_t<2>(g, st);
    }
}

hipError_t launch_gemm_nt(const GemmArgs& a, int mode, bool aligned, hipStream_t st) {
    if (sim_strip_eligible(a, mode, aligned)) return launch_sim_strip(a, mode, st);     // K = 512 similarity: the strip kernel
    switch (mode) {
        case GEMM_F32: return launch_m<GEMM_F32>(a, aligned, st);
        case GEMM_F16: return launch_m<GEMM_F16>(a, aligned, st);
        case GEMM_BF16: return launch_m<GEMM_BF16>(a, aligned, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace laff
