// fc_strip.hip -- TransformNet.forward (FC -> activation -> folded BatchNorm; /root/reference/model/model.py:257-276) for K = 512 inputs
// as a STRIP kernel for gfx950, X stationary:
//
//   Y[r][c] = bn_s[c] * act( sum_k X[r][k] W[c][k] + bias[c] ) + bn_t[c]          fp32 in / fp32 out, fp32-class accuracy
//
// on the fp16 matrix pipe: X = s_r (Xhi + Xlo), W = s_c (Whi + Wlo) with power-of-two scales (row maximum scaled into [512, 1024)), the
// three products lo*hi + hi*lo + hi*hi accumulate in fp32 (fp16 x fp16 products are exact in fp32), the epilogue undoes the scales.
//
// The tiled kernel this replaces (gemm_nt.hip, gemm_tile_x3 with the input split fused) ran at 39 % MFMA utilisation and needed a separate
// pass over the inputs for the row scales.  Here:
//   * a workgroup is 4 wavefronts, ONE PER SIMD, 512 registers each.  A wavefront owns 32 input rows: it loads them ONCE (fp32, straight
//     into the accumulator half of the register file), finds the row maxima there (a row lives in two lanes), and converts them in
//     place into the MFMA A fragments of hi and lo for the WHOLE K = 512 (2 x 128 registers).  No row-scale launch, no split planes
//     in memory, the conversion happens once per row -- not once per column tile;
//   * only W moves: it is packed once per model (laff_fc_strip_pack) into an LDS image -- per block of 32 output columns and half of
//     K one 32 KiB ring slot {hi plane, lo plane} x 16 sub-steps x 64 lanes x 16 bytes, i.e. exactly the ds_read_b128 fragment
//     order -- so the LDS-DMA is a linear copy and a fragment read is lane-linear (conflict-free by construction).  W of a feature is
//     1 MiB: it streams from L2;
//   * per sub-step (16 k) two fragment reads feed three MFMAs 32x32x16 (Xlo.Whi, Xhi.Wlo, Xhi.Whi);
//   * MFMA roles: A = X rows, B = W columns, so a lane's accumulators are 16 ROWS of ONE output column: the per-column epilogue
//     parameters are four lane constants per block, and a store instruction (dword per lane) writes 2 rows x 128 contiguous bytes --
//     whole cache lines without a transposing slab;
//   * two accumulator sets alternate: the epilogue of block b - 1 (scale, bias, tanh on v_exp/v_rcp, BatchNorm, store) is issued in
//     the shadow of block b's MFMAs, a few instructions behind each one, at places computed at compile time (make_plan) together
//     with the operands of the counted s_waitcnt;
//   * persistent grid: the (feature, strip, column block) units of ALL problems of a launch are cut into equal contiguous ranges.
//
// The k index inside a sub-step is permuted (any bijection is legal as long as A and B agree): lane (i, hh) element e holds
// k = 16 j + 4 hh + (e & 3) + 8 (e >> 2), so that the strip load's 16-byte pieces of lanes hh = 0, 1 are adjacent in the row.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "kernels.h"
#include "strip_util.h"

namespace laff {

namespace {

using namespace su;

typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int FR = FC_STRIP_ROWS;              // 128 rows per strip: 4 wavefronts x 32
constexpr int SLOT = 32 * 1024;                // one ring slot: 32 columns x 256 k x {hi, lo}
constexpr int PLANE = 16 * 1024;
constexpr int RING = 4;
constexpr int PIECES = 8;                      // 1 KiB LDS-DMA pieces per wave per slot
constexpr int XB_OFF = RING * SLOT;            // a fifth 32 KiB buffer, strip staging only (all of the CU's LDS: 160 KiB)
constexpr int SMEM = XB_OFF + SLOT;
static_assert(SMEM <= 160 * 1024, "LDS budget");

constexpr int FD = 6;                          // fragment reads run FD sub-steps ahead,
constexpr int NSETS = 8;                       // into 8 register sets (two planes each; 32 sub-steps per block: a divisor of 32)
static_assert(32 % NSETS == 0 && FD < NSETS, "the set of a sub-step must not depend on the block");
constexpr int BAR_J = 16 - FD;                 // the body's barrier sits in front of this sub-step
constexpr int NSLOT = 96;                      // MFMA issue slots of a column block: 2 bodies x 16 sub-steps x 3

// ---- all 256 accumulator registers are named literally (the strip).  hipcc must not put anything of its own there (left alone it parks
// long-lived values in a0.. when the 256 architectural registers get tight -- silently overwritten by the strip): 64 dummy quads are
// defined in the accumulator file at kernel entry and "used" at its exit, so the allocator sees every one of them occupied throughout;
// if registers run out it spills to scratch instead, which tools/debug/isa_audit.py reports. ----
struct AgprHold { u32x4 q[64]; };
#define HOLD16(OP, h, b)                                                                                                                  \
    OP(h.q[b], h.q[b + 1], h.q[b + 2], h.q[b + 3], h.q[b + 4], h.q[b + 5], h.q[b + 6], h.q[b + 7], h.q[b + 8], h.q[b + 9], h.q[b + 10], h.q[b + 11], \
       h.q[b + 12], h.q[b + 13], h.q[b + 14], h.q[b + 15])
#define HOLD_DEF(x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, xa, xb, xc, xd, xe, xf)                                                            \
    asm volatile("" : "=a"(x0), "=a"(x1), "=a"(x2), "=a"(x3), "=a"(x4), "=a"(x5), "=a"(x6), "=a"(x7), "=a"(x8), "=a"(x9), "=a"(xa), "=a"(xb), \
                 "=a"(xc), "=a"(xd), "=a"(xe), "=a"(xf))
#define HOLD_USE(x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, xa, xb, xc, xd, xe, xf)                                                            \
    asm volatile("" ::"a"(x0), "a"(x1), "a"(x2), "a"(x3), "a"(x4), "a"(x5), "a"(x6), "a"(x7), "a"(x8), "a"(x9), "a"(xa), "a"(xb), "a"(xc),   \
                 "a"(xd), "a"(xe), "a"(xf))
__device__ __forceinline__ void agpr_hold_begin(AgprHold& h) { HOLD16(HOLD_DEF, h, 0); HOLD16(HOLD_DEF, h, 16); HOLD16(HOLD_DEF, h, 32); HOLD16(HOLD_DEF, h, 48); }
__device__ __forceinline__ void agpr_hold_end(AgprHold& h) { HOLD16(HOLD_USE, h, 0); HOLD16(HOLD_USE, h, 16); HOLD16(HOLD_USE, h, 32); HOLD16(HOLD_USE, h, 48); }

template <int R, int OFF>
__device__ __forceinline__ void agpr_load4(const void* p) {            // a[R .. R + 3] <- 16 bytes at p + OFF
    asm volatile("global_load_dwordx4 a[%c1:%c2], %0, off offset:%c3" ::"v"(p), "n"(R), "n"(R + 3), "n"(OFF) : "memory");
}
template <int R>
__device__ __forceinline__ float agpr_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(R));
    return x;
}
template <int R>
__device__ __forceinline__ void agpr_write(unsigned x) { asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(x), "n"(R)); }

// acc (+)= A(strip fragment in a[R .. R + 3]: rows of X) x B(fragment of the W stream: output columns)
template <int R, bool FIRST>
__device__ __forceinline__ void mfma_strip(f32x16& acc, const u32x4& wfrag) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c2:%c3], %1, 0" : "=v"(acc) : "v"(wfrag), "n"(R), "n"(R + 3));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c2:%c3], %1, %0" : "+v"(acc) : "v"(wfrag), "n"(R), "n"(R + 3));
}

template <int IMM>
__device__ __forceinline__ void lds_read128(u32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}

// one 1 KiB piece of the W stream, straight into LDS (lane L lands at M0 base + 16 L; the image is already in LDS order)
template <int LDSOFF>
__device__ __forceinline__ void dma_piece(unsigned voff, u32x4 rsrc, unsigned soff, unsigned m0base) {
    asm volatile("s_add_i32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 ::"v"(voff), "s"(rsrc), "s"(soff), "s"(m0base), "n"(LDSOFF) : "memory", "scc");
}

// 1 KiB of input rows straight into LDS (lane L lands at M0 base + LDSOFF + 16 L): scalar base + per-lane 32-bit offset.  (No
// immediate offset: the instruction's offset field moves the LDS address as well as the global one.)
template <int LDSOFF>
__device__ __forceinline__ void dma_rows(unsigned voff, unsigned long long base, unsigned m0base) {
#ifndef LAFF_FCS_XFLAVOR
#define LAFF_FCS_XFLAVOR ""
#endif
    asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 " LAFF_FCS_XFLAVOR
                 ::"v"(voff), "s"(base), "s"(m0base), "n"(LDSOFF) : "memory", "scc");
}
// 16 bytes of the LDS staging -> a[R .. R + 3]
template <int R, int IMM>
__device__ __forceinline__ void lds_to_agpr(unsigned addr) {
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "n"(R), "n"(R + 3), "n"(IMM) : "memory");
}
template <typename T>
__device__ __forceinline__ void pin_v(T& x) { asm volatile("" : "+v"(x)); }
// address = voff0 + ROWMUL * ldy4, made right in front of the store (one address register instead of one per accumulator)
template <int ROWMUL>
__device__ __forceinline__ void store_row(unsigned& tmp, float data, unsigned voff0, unsigned ldy4, u32x4 rsrc, unsigned soff) {
#ifndef LAFF_FCS_STFLAVOR
#define LAFF_FCS_STFLAVOR "nt"
#endif
    asm volatile("v_mad_u32_u24 %0, %3, %4, %2\n\tbuffer_store_dword %1, %0, %5, %6 offen " LAFF_FCS_STFLAVOR
                 : "=&v"(tmp) : "v"(data), "v"(voff0), "s"(ldy4), "n"(ROWMUL), "s"(rsrc), "s"(soff) : "memory");
}

// row maximum -> exponent of the power-of-two scale: same rule as split_rows_kernel (fuse.hip): maximum scaled into [512, 1024)
__device__ __forceinline__ int split_exponent(float m) {
    const int be = (int)((__float_as_uint(m) >> 23) & 0xffu);
    if (!(m > 0.f) || be == 0xff) return 0;
    return max(be - 127, -100);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
// one raw chunk (two quads from the staging read-back) -> a[R .. R + 7], its eight values folded into the two running row maxima
template <int R>
__device__ __forceinline__ void stash_chunk(const u32x4& q0, const u32x4& q1, float& m0, float& m1) {
    asm volatile("v_max3_f32 %0, |%2|, |%3|, %0\n\tv_max3_f32 %1, |%4|, |%5|, %1\n\tv_max3_f32 %0, |%6|, |%7|, %0\n\tv_max3_f32 %1, |%8|, |%9|, %1\n\t"
                 "v_accvgpr_write_b32 a[%c10+0], %2\n\tv_accvgpr_write_b32 a[%c10+1], %3\n\tv_accvgpr_write_b32 a[%c10+2], %4\n\t"
                 "v_accvgpr_write_b32 a[%c10+3], %5\n\tv_accvgpr_write_b32 a[%c10+4], %6\n\tv_accvgpr_write_b32 a[%c10+5], %7\n\t"
                 "v_accvgpr_write_b32 a[%c10+6], %8\n\tv_accvgpr_write_b32 a[%c10+7], %9"
                 : "+v"(m0), "+v"(m1) : "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w), "n"(R));
}
// a[R .. R + 7] (eight fp32: k = 16 g + 4 hh + {0..3}, + 8) -> a[R .. R + 3] = their fp16 hi halves, a[R + 4 .. R + 7] = the lo halves.
//   hi = f16(x s) (x s is exact: s is a power of two), residual x s - hi exact in fp32, lo = f16(residual): split_rows_kernel's
//   arithmetic (fuse.hip), as one statement with the four pairs' chains interleaved (a dependent VALU pair costs 7 cycles, not 4)
template <int R>
__device__ __forceinline__ void convert_chunk_inplace(float s) {
    float x0, x1, x2, x3, x4, x5, x6, x7;
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    asm volatile(
        "v_accvgpr_read_b32 %0, a[%c17+0]\n\tv_accvgpr_read_b32 %1, a[%c17+1]\n\tv_accvgpr_read_b32 %2, a[%c17+2]\n\tv_accvgpr_read_b32 %3, a[%c17+3]\n\t"
        "v_accvgpr_read_b32 %4, a[%c17+4]\n\tv_accvgpr_read_b32 %5, a[%c17+5]\n\tv_accvgpr_read_b32 %6, a[%c17+6]\n\tv_accvgpr_read_b32 %7, a[%c17+7]\n\t"
        "v_fma_mixlo_f16 %8, %0, %16, 0\n\tv_fma_mixlo_f16 %9, %2, %16, 0\n\tv_fma_mixlo_f16 %10, %4, %16, 0\n\tv_fma_mixlo_f16 %11, %6, %16, 0\n\t"
        "v_fma_mixhi_f16 %8, %1, %16, 0\n\tv_fma_mixhi_f16 %9, %3, %16, 0\n\tv_fma_mixhi_f16 %10, %5, %16, 0\n\tv_fma_mixhi_f16 %11, %7, %16, 0\n\t"
        "v_fma_mix_f32 %0, %0, %16, -%8 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %2, %2, %16, -%9 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %4, %16, -%10 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %6, %6, %16, -%11 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %1, %1, %16, -%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %3, %3, %16, -%9 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %5, %5, %16, -%10 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %7, %7, %16, -%11 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_accvgpr_write_b32 a[%c17+0], %8\n\tv_accvgpr_write_b32 a[%c17+1], %9\n\tv_accvgpr_write_b32 a[%c17+2], %10\n\tv_accvgpr_write_b32 a[%c17+3], %11\n\t"
        "v_cvt_pk_f16_f32 %12, %0, %1\n\tv_cvt_pk_f16_f32 %13, %2, %3\n\tv_cvt_pk_f16_f32 %14, %4, %5\n\tv_cvt_pk_f16_f32 %15, %6, %7\n\t"
        "v_accvgpr_write_b32 a[%c17+4], %12\n\tv_accvgpr_write_b32 a[%c17+5], %13\n\tv_accvgpr_write_b32 a[%c17+6], %14\n\tv_accvgpr_write_b32 a[%c17+7], %15"
        : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7), "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3),
          "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(s), "n"(R));
}
__device__ __forceinline__ void absmax3(float& m, float x, float y) { asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(x), "v"(y)); }
// hi = f16(x s) for two values (x s is exact: s is a power of two), the residuals x s - hi (exact in fp32) replace x, lo = f16(residual):
// the arithmetic of split_rows_kernel (fuse.hip)
__device__ __forceinline__ void split_pair(float& x0, float& x1, float s, unsigned& hi, unsigned& lo) {
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(x0), "v"(s));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "v"(s));
    asm volatile("v_fma_mix_f32 %0, %0, %1, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(x0) : "v"(s), "v"(hi));
    asm volatile("v_fma_mix_f32 %0, %0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(x1) : "v"(s), "v"(hi));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(x0), "v"(x1));
}

// ---- the epilogue of a column block as a static stream of micro-ops -----------------------------------------------------------------
// Per accumulator i (row 8 (i >> 2) + 4 hh + (i & 3) of the wave's 32, the lane's column):
//   MUL t = acc * rs[i]   FMA1 t = t * c.x + c.y   [EXP t = 2^t   ADD1 t = t + 1   RCP t = 1 / t] | [MAX t = max(t, 0)]   FMA2 t = t * c.z + c.w   ST
// with the lane constants c = {cs', b', c1, c0} folded by laff_fc_strip_pack so that the chain is the whole FC epilogue:
//   tanh:    y' = 2 log2(e) (cs acc rs + bias), tanh = 1 - 2 / (2^y' + 1), out = (bn_s + bn_t) + (-2 bn_s) / (2^y' + 1)
//   sigmoid: y' = -log2(e) (...),              out = bn_t + bn_s / (2^y' + 1)
//   relu:    out = bn_s max(y, 0) + bn_t;      none: out = (cs bn_s) acc rs + (bn_s bias + bn_t)   (one fma)
// Elements go in groups of four, stage by stage, so that a dependent pair is four instructions apart (the transcendental ->
// VALU forwarding hazard needs one; hipcc pads nothing inside or between asm statements it cannot see into).
enum : unsigned char { OP_WAITCV = 0, OP_MUL, OP_FMA1, OP_EXP, OP_ADD1, OP_RCP, OP_MAX, OP_FMA2, OP_ST };
enum { ACT_LIN = 0, ACT_RELU = 1, ACT_EXP = 2 };                 // epilogue kinds (template parameter)
struct EpiOp { unsigned char kind, elem; };
struct EpiStream { EpiOp op[160]; int n; };
constexpr int op_cost(unsigned char k) { return (k == OP_EXP || k == OP_RCP) ? 3 : (k == OP_WAITCV ? 0 : (k == OP_ST ? 2 : 1)); }

template <int ACTK>
constexpr EpiStream make_stream() {
    EpiStream s{};
    s.n = 0;
    auto push = [&](unsigned char k, int e) { s.op[s.n++] = EpiOp{k, (unsigned char)e}; };
    push(OP_WAITCV, 0);
    for (int g = 0; g < 4; ++g) {
        auto stage = [&](unsigned char k) { for (int e = 0; e < 4; ++e) push(k, 4 * g + e); };
        stage(OP_MUL);
        stage(OP_FMA1);
        if (ACTK == ACT_EXP) { stage(OP_EXP); stage(OP_ADD1); stage(OP_RCP); stage(OP_FMA2); }
        if (ACTK == ACT_RELU) { stage(OP_MAX); stage(OP_FMA2); }
        stage(OP_ST);
    }
    return s;
}

// ---- the schedule of one column block (two bodies H = 0, 1 of 16 sub-steps J, three MFMAs M each): slot sg = 48 H + 3 J + M ----------
//   * (J, 0): the two fragment reads (hi, lo plane) of the sub-step FD ahead -- from J = BAR_J on that is the NEXT ring slot;
//   * in front of sub-step BAR_J: counted vmcnt (this wave's pieces of the next slot have landed) + s_barrier (everybody's have, and
//     everybody is done with the previous slot, which the pieces issued right behind the barrier refill -- 4 slots: 3 ahead);
//   * (BAR_J .. 15, M = 1) and two M = 2 slots: the 8 DMA pieces;  (H = 0, J = 0, M = 1): this block's four lane constants (one load);
//     -- in the LAST THREE bodies of a segment those pieces would fetch W slots beyond it: they carry the first three K-eighth rounds
//     of the NEXT strip's input rows instead (same count, same LDS targets: this wave's 8 KiB of the slot being freed); the first of
//     the three also issues round 3 into the staging-only fifth buffer, and round 4 goes into the last body's slot in front of the
//     drain: the strip switch starts with 40 of its 64 KiB per wave in LDS or on the way;
//   * everything else: the epilogue stream of the previous block, spread evenly (cost-weighted).  It starts behind the third MFMA
//     of the block (the previous block's last MFMA has retired by then) and ends before the block does.
constexpr int slot_piece(int sg) {
    const int J = (sg % 48) / 3, M = sg % 3;
    if (M == 1 && J >= BAR_J) return J - BAR_J;          // one per sub-step behind the barrier ...
    if (M == 2 && J == BAR_J + 1) return 6;              // ... and two sub-steps with a second one: the four waves leave the barrier
    if (M == 2 && J == BAR_J + 4) return 7;              // together, and pieces packed into four sub-steps queue up in the CU's one TA path
    return -1;
}
constexpr bool slot_cvload(int sg) { return sg == 1; }
struct Plan {
    short begin[NSLOT + 1];
    short vm_bar[2];      // vmcnt operand at the barrier of body H
    short vm_cv;          // vmcnt operand in front of the first use of the previous block's lane constants
    bool fits;
};
template <int ACTK>
constexpr Plan make_plan() {
    Plan p{};
    const EpiStream st = make_stream<ACTK>();
    int usable = 0, total = 0;
    auto ok = [](int sg) { return sg >= 3 && slot_piece(sg) < 0; };
    for (int sg = 0; sg < NSLOT; ++sg) usable += ok(sg) ? 1 : 0;
    for (int i = 0; i < st.n; ++i) total += op_cost(st.op[i].kind);
    usable -= 4;                                        // finish a few slots early
    int at = 0, k = 0, spent = 0;
    for (int sg = 0; sg < NSLOT; ++sg) {
        p.begin[sg] = (short)at;
        if (!ok(sg)) continue;
        ++k;
        const int target = (int)(((long)k * total + usable - 1) / usable);
        while (at < st.n && spent < target) { spent += op_cost(st.op[at].kind); ++at; }
    }
    p.begin[NSLOT] = (short)at;
    p.fits = at == st.n;
    // vector-memory program order over three consecutive identical blocks; the third one is measured
    int vm_n = 0, last_piece[3][2] = {}, cv_ord[3] = {}, bar_at[3][2] = {}, waitcv_at[3] = {};
    for (int blk = 0; blk < 3; ++blk)
        for (int sg = 0; sg < NSLOT; ++sg) {
            const int H = sg / 48, J = (sg % 48) / 3, M = sg % 3;
            if (J == BAR_J && M == 0) bar_at[blk][H] = vm_n;
            if (slot_cvload(sg)) cv_ord[blk] = ++vm_n;
            if (slot_piece(sg) >= 0) { ++vm_n; last_piece[blk][H] = vm_n; }
            for (int i = p.begin[sg]; i < p.begin[sg + 1]; ++i) {
                if (st.op[i].kind == OP_WAITCV) waitcv_at[blk] = vm_n;
                if (st.op[i].kind == OP_ST) ++vm_n;
            }
        }
    auto clamp = [](int c) { return c < 0 ? 0 : (c > 63 ? 63 : c); };
    // barrier of body (b, H): the pieces of ring slot sigma + 1 were issued in body sigma - 2 = (b - 1, H)
    for (int H = 0; H < 2; ++H) p.vm_bar[H] = (short)clamp(bar_at[2][H] - last_piece[1][H]);
    p.vm_cv = (short)clamp(waitcv_at[2] - cv_ord[1]);
    return p;
}

}  // namespace

template <int ACTK>
__global__ __launch_bounds__(256, 1) void fc_strip_kernel(const FcStripArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    AgprHold hold;
    agpr_hold_begin(hold);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n31 = lane & 31, hh = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);

    const int nblk = pin_s(a.nblk), count = pin_s(a.count);
    const int nranges = pin_s(a.nranges);
    const long U = (long)pin_s((unsigned)a.total_units);
    const int myrange = a.range_of_wg[blockIdx.x];
    int u0 = pin_s((int)(U * myrange / nranges));              // (U < 2^31: launch_fc_strip)
    const int u1 = pin_s((int)(U * (myrange + 1) / nranges));
    const unsigned img_bytes = (unsigned)nblk * (2u * SLOT);

    constexpr Plan PLAN = make_plan<ACTK>();
    constexpr EpiStream STREAM = make_stream<ACTK>();
    constexpr int DRAIN_STORES = 16;                      // OP_ST items of the stream: what a drain leaves in the vector-memory queue
    static_assert(PLAN.fits, "the epilogue stream does not fit behind the MFMAs of one column block");

    // per-lane constants of the loops
    unsigned lane16 = (unsigned)lane * 16u;
    asm volatile("" : "+v"(lane16));
    const unsigned xa0 = lds0 + lane16, xa1 = lds0 + lane16 + 2u * SLOT;      // fragment addresses: ring slots 0, 1 | 2, 3
    const unsigned wslot = (unsigned)wave * (PIECES * 1024u);                   // this wave's 8 KiB of a slot (LDS and source offset)
    const unsigned ldsw = pin_s(lds0 + wslot);                                  // LDS address of this wave's 8 KiB of ring slot 0
    const unsigned cvoff = (unsigned)n31 * 16u;
    int pre_base = -1;           // >= 0: the previous segment's last three bodies brought rounds 0..2 of this segment's strip into buffers (e + pre_base) & 3
    const unsigned m0_keep = m0_get();
#ifdef LAFF_FCS_TRACE
    // debug build: cycle stamps of wave 0 -- per segment s (up to 8): base 8 s: +0 start, +1 strip in the registers (raw), +2 row maxima,
    // +3 converted, +4 ring prologue landed + barrier, +5 block loop done, +6 drained + segment end | [64 + s] = blocks of the segment
    unsigned long long* const trc = a.trace ? a.trace + (size_t)blockIdx.x * 80 : nullptr;
    int trc_seg = 0;
#define STAMP(i) do { if (trc && tid == 0 && trc_seg < 8) trc[8 * trc_seg + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

    while (u0 < u1) {
        // ---- the segment: column blocks [blk0, blk0 + n) of one strip of one problem --------------------------------------------------
        int p = 0;
        while (p + 1 < count && u0 >= a.p[p + 1].unit0) ++p;
        p = __builtin_amdgcn_readfirstlane(p);
        const int ul = u0 - a.p[p].unit0;
        const int strip = __builtin_amdgcn_readfirstlane(ul / nblk);
        const int blk0 = __builtin_amdgcn_readfirstlane(ul - strip * nblk);
        const int n = __builtin_amdgcn_readfirstlane(std::min(nblk - blk0, u1 - u0));
        u0 += n;
        STAMP(0);
#ifdef LAFF_FCS_TRACE
        if (trc && tid == 0 && trc_seg < 8) trc[64 + trc_seg] = (unsigned long long)n;
#endif
        // Input rows travel in K-eighth ROUNDS: 32 rows x 256 bytes per wave = 8 LDS-DMA instructions of 4 rows x 256 contiguous bytes
        // (direct 16-byte loads in the fragment layout touch 32 rows per instruction and run at a fifth of the rate).  Instruction t, lane
        // L: row 4 t + (L >> 4) of the wave's 32, LDS position L & 15 of that row <- source piece (L & 15) ^ (row & 15): the swizzle makes
        // the fragment-order read-back conflict-free.  xoff(p, strip, xv): the per-lane source offsets (rows beyond the matrix read the
        // last row), relative to the strip's first row.
        auto xoff = [&](int p_, int strip_, unsigned (&xv)[8]) {
            const int N_ = a.p[p_].N, ldx_ = a.p[p_].ldx, r0_ = strip_ * FR;
            int ln = lane;
            asm volatile("" : "+v"(ln));              // made here: hipcc hoists lane-only terms out of the segment loop and keeps them all the way
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int rw = 4 * t + (ln >> 4);                       // row of the wave
                const int r = std::min(r0_ + wave * 32 + rw, N_ - 1) - r0_;
                xv[t] = (unsigned)r * (unsigned)ldx_ * 4u + ((((unsigned)ln & 15u) ^ ((unsigned)rw & 15u)) << 4);
            }
        };
        // the strip whose first three rounds this segment's last three bodies bring in: the next segment's (n >= 2: three bodies exist)
#ifdef LAFF_FCS_NOEARLY
        const bool has_next = false;
#else
        const bool has_next = u0 < u1 && n >= 2;
#endif
        int p2 = p, strip2 = strip;
        if (has_next) {
            p2 = 0;
            while (p2 + 1 < count && u0 >= a.p[p2 + 1].unit0) ++p2;
            p2 = __builtin_amdgcn_readfirstlane(p2);
            strip2 = __builtin_amdgcn_readfirstlane((u0 - a.p[p2].unit0) / nblk);
        }
        unsigned xvn[8];
        xoff(p2, strip2, xvn);
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(xvn[t]));
        const u32x4 rsrcXn = rebased_rsrc((unsigned long long)a.p[p2].X + (unsigned long long)(strip2 * FR) * (unsigned)a.p[p2].ldx * 4ull,
                                          (unsigned long long)FR * (unsigned)a.p[p2].ldx * 4ull, 0ull);
        const int N = pin_s(a.p[p].N), ldx = pin_s(a.p[p].ldx), ldy = pin_s(a.p[p].ldy);
        const unsigned long long pX = pin_s((unsigned long long)a.p[p].X), pW = pin_s((unsigned long long)a.p[p].img);
        const unsigned long long pVec = pin_s((unsigned long long)a.p[p].vec), pY = pin_s((unsigned long long)a.p[p].Y);
        const int row0 = strip * FR;

        f32x16 acc[2];
        u32x4 fr[NSETS][2];
        float rs[16];
        float tt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned ta[4] = {0, 0, 0, 0};
        f32x4 cv[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};

        int e_row = 0;                   // exponent of this lane's row maximum: scale 2^(9 - e) in, 2^(e - 9) out
        // ---- the strip: this wave's 32 rows x 512 fp32 -> a[0:255] (raw), eight rounds through four wave-private buffers -- this wave's
        // 8 KiB of each (idle) ring slot; round e lives in buffer (e + base) & 3.  Rounds 0..2 are already there when the previous
        // segment's last bodies brought them (pre_base); up to four rounds are in flight.  The read-back (lane (row n31, half hh) takes
        // 16-byte pieces 4 cc + hh and 4 cc + 2 + hh of chunk cc) folds the row maxima and parks the raw values in the accumulator file.
        {
            const unsigned long long xbase = pX + (unsigned long long)row0 * (unsigned)ldx * 4ull;
            unsigned xv[8];
            xoff(p, strip, xv);
            const bool pre = pre_base >= 0;
            const int base = pre ? pre_base : 0;
            unsigned XA[8];                            // read-back addresses inside a buffer: (cc, quad) -> piece 4 cc + 2 quad + hh, swizzled
            {
                unsigned ln = (unsigned)lane;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    XA[j] = (ln & 31u) * 256u + ((((unsigned)(4 * (j >> 1) + 2 * (j & 1)) + (ln >> 5)) ^ (ln & 15u)) << 4);
            }
            // round -> buffer.  Without early rounds: 0 1 2 3 X 0 1 2 (five in flight).  With them: rounds 0..2 sit in ring slots b0 b1 b2
            // (b_i = (i + base) & 3), round 3 in X, round 4 in b3 (the previous segment's last slot): b0 b1 b2 X b3 b0 b1 X -- round 7 follows
            // round 3, which landed long ago, not round 2, the last one to have been issued.
            auto buf_of = [&](int e) {
                const int ring = pre ? (e < 3 ? e : (e == 4 ? 3 : e - 5)) : (e < 4 ? e : e - 5);
                const bool x = pre ? (e == 3 || e == 7) : e == 4;
                return ldsw + (unsigned)(x ? RING : ((ring + base) & 3)) * (unsigned)SLOT;
            };
            auto issue_round = [&](auto EC) {
                constexpr int e = decltype(EC)::value;
                const unsigned long long ebase = xbase + (unsigned long long)(e * 256);
                const unsigned b = pin_s(buf_of(e));
                static_for<0, 8>([&](auto TC) {
                    constexpr int t = decltype(TC)::value;
                    dma_rows<t * 1024>(xv[t], ebase, b);
                });
            };
            float m0 = 0.f, m1 = 0.f;                  // running maxima of |x| over this lane's half of its row
            u32x4 rq[8];
            auto read_round = [&](auto EC) {
                constexpr int e = decltype(EC)::value;
                const unsigned b = buf_of(e);
                static_for<0, 8>([&](auto JC) { constexpr int j = decltype(JC)::value; lds_read128<0>(rq[j], XA[j] + b); });
                wait_lgkm<0>();
                static_for<0, 8>([&](auto IC) { pin_v(rq[decltype(IC)::value]); });
            };
            auto stash_round = [&](auto EC) {
                constexpr int e = decltype(EC)::value;
                static_for<0, 4>([&](auto CC) {
                    constexpr int cc = decltype(CC)::value;
                    stash_chunk<8 * (4 * e + cc)>(rq[2 * cc], rq[2 * cc + 1], m0, m1);
                });
            };
            using E0 = std::integral_constant<int, 0>; using E1 = std::integral_constant<int, 1>; using E2 = std::integral_constant<int, 2>;
            using E3 = std::integral_constant<int, 3>; using E4 = std::integral_constant<int, 4>; using E5 = std::integral_constant<int, 5>;
            using E6 = std::integral_constant<int, 6>; using E7 = std::integral_constant<int, 7>;
            if (!pre) {
                issue_round(E0{}); issue_round(E1{}); issue_round(E2{}); issue_round(E3{}); issue_round(E4{});
                wait_vm<32>(); read_round(E0{}); issue_round(E5{}); stash_round(E0{});
                wait_vm<32>(); read_round(E1{}); issue_round(E6{}); stash_round(E1{});
                wait_vm<32>(); read_round(E2{}); issue_round(E7{}); stash_round(E2{});
                wait_vm<32>(); read_round(E3{}); stash_round(E3{});
                wait_vm<24>(); read_round(E4{}); stash_round(E4{});
            } else {
                // The queue holds, oldest first: round 3 and round 0 (tail body 1), round 1 (body 2: 8 pieces, 8 stores, the lane constants),
                // round 2 (body 3: 8 + 8), round 4 (8, in front of the drain), the drain's stores.  The counts are lower bounds of what was
                // issued behind the round waited for.
                constexpr int DR = DRAIN_STORES;
                wait_vm<17 + 16 + 8 + DR>(); read_round(E0{}); issue_round(E5{}); stash_round(E0{});
                wait_vm<16 + 8 + DR + 8>(); read_round(E1{}); issue_round(E6{}); stash_round(E1{});
                read_round(E3{}); issue_round(E7{}); stash_round(E3{});
                wait_vm<8 + DR + 24>(); read_round(E2{}); stash_round(E2{});
                wait_vm<24>(); read_round(E4{}); stash_round(E4{});
            }
            wait_vm<16>(); read_round(E5{}); stash_round(E5{});
            wait_vm<8>(); read_round(E6{}); stash_round(E6{});
            wait_vm<0>(); read_round(E7{}); stash_round(E7{});
            float m = fmaxf(m0, m1);
            m = fmaxf(m, __shfl_xor(m, 32));           // a row lives in lanes l and l + 32
            e_row = split_exponent(m);
            pre_base = has_next ? ((n & 1) ? 2 : 0) : -1;   // where this segment's last three bodies will leave the next strip's rounds 0..2
        }
        __builtin_amdgcn_s_barrier();            // every wave has emptied its staging quarter: the ring may fill
        asm volatile("" ::: "memory");
        STAMP(1);

        // ---- W ring prologue: the first three slots of the segment (slot s of the segment = image slot 2 blk0 + s); it lands while the
        // strip is converted.  The image as a raw buffer starting at this wave's 8 KiB of a slot ----
        const u32x4 rsrcW = rebased_rsrc(pW, img_bytes, wslot);
        unsigned soffW = (unsigned)blk0 * (2u * SLOT);                        // image offset of the slot the next pieces belong to
        static_for<0, 3>([&](auto SC) {
            constexpr int s = decltype(SC)::value;
            static_for<0, PIECES>([&](auto PC) {
                constexpr int P = decltype(PC)::value;
                dma_piece<s * SLOT + P * 1024>(lane16 + (unsigned)(P * 1024), rsrcW, std::min(soffW, img_bytes - SLOT), ldsw);
            });
            soffW += SLOT;
        });

        // ---- the raw strip is converted in place ----
        STAMP(2);
        {
            const float s = pow2f(9 - e_row);
            asm volatile("s_nop 1" ::: "memory");
            // sub-step g: hi <- a[8 g .. 8 g + 3], lo <- a[8 g + 4 .. 8 g + 7]; halves e = 0 .. 7 <-> raw registers 8 g + e
            static_for<0, 32>([&](auto GC) { convert_chunk_inplace<8 * decltype(GC)::value>(s); });
        }
        asm volatile("s_nop 3" ::: "memory");            // v_accvgpr_write -> MFMA reading it
        STAMP(3);
        {
            // the epilogue's row scales: lane (n31, hh) needs those of rows 8 q + 4 hh + e, which lane 8 q + 4 hh + e has
            const float mine = pow2f(e_row - 9);
#pragma unroll
            for (int i = 0; i < 16; ++i) rs[i] = __shfl(mine, 8 * (i >> 2) + 4 * hh + (i & 3));
        }
        // output addressing: row 8 q + 4 hh + e of the wave's 32 at voff0 + (8 q + e) ldy4
        const unsigned ldy4 = (unsigned)ldy * 4u;
        unsigned voff0 = ((unsigned)(wave * 32 + 4 * hh) * (unsigned)ldy + (unsigned)n31) * 4u;
        // the output rows of this strip as a raw buffer: rows beyond N are dropped by its bounds check (voff carries the row)
        const u32x4 rsrcY = rebased_rsrc(pY + (unsigned long long)row0 * (unsigned)ldy * 4ull,
                                         ((unsigned long long)(std::min(FR, N - row0) - 1) * (unsigned)ldy + (unsigned)(nblk * 32)) * 4ull, 0ull);
        const u32x4 rsrcNone = {0u, 0u, 0u, 0x00020000u};
        // the lane constants {cs', b', c1, c0} of column 32 blk + n31: [D][4] floats
        const u32x4 rsrcV = rebased_rsrc(pVec, (unsigned long long)nblk * 512ull, 0ull);

        // everything the prologue loaded is in registers by now: pin hipcc's own waits here, not inside the hand-counted loops
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(rs[i]));
        asm volatile("" : "+v"(voff0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(4);
        // fragments of sub-steps 0 .. FD - 1 of the first slot
        static_for<0, FD>([&](auto JC) {
            constexpr int j = decltype(JC)::value;
            lds_read128<j * 1024>(fr[j][0], xa0);
            lds_read128<PLANE + j * 1024>(fr[j][1], xa0);
        });

        // ---- one micro-op of the epilogue of block `blk - 1` (accumulator set Q, lane constants cv[Q]) --------------------------------
        u32x4 rsrcYe = rsrcNone;            // the first body's epilogue runs on nothing: an empty buffer drops its stores
        unsigned soffY = 0u;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using IM1 = std::integral_constant<int, -1>;
        auto epi_item = [&](auto QC, auto IC, auto DRAIN) {
            constexpr int Q = decltype(QC)::value;
            constexpr EpiOp op = STREAM.op[decltype(IC)::value];
            constexpr int i = op.elem;
            if constexpr (op.kind == OP_WAITCV) {
                // (drain: the lane constants were the block's first vector-memory operation: 16 pieces and 16 stores came behind them --
                // a vmcnt(0) here would also wait for the next strip's rows the last bodies have just asked for)
                // (a drain: the two bodies behind the load issued 2 x (8 pieces + 8 stores); DRAIN counts what else was)
                if constexpr (decltype(DRAIN)::value >= 0) wait_vm<32 + decltype(DRAIN)::value>(); else wait_vm<PLAN.vm_cv>();
                asm volatile("" : "+v"(cv[Q]));
            } else if constexpr (op.kind == OP_MUL) {
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tt[i & 7]) : "v"(acc[Q][i]), "v"(rs[i]));
            } else if constexpr (op.kind == OP_FMA1) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(tt[i & 7]) : "v"(cv[Q].x), "v"(cv[Q].y));
            } else if constexpr (op.kind == OP_EXP) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(tt[i & 7]));
            } else if constexpr (op.kind == OP_ADD1) {
                asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(tt[i & 7]));
            } else if constexpr (op.kind == OP_RCP) {
                asm volatile("v_rcp_f32 %0, %0" : "+v"(tt[i & 7]));
            } else if constexpr (op.kind == OP_MAX) {
                asm volatile("v_max_f32 %0, 0, %0" : "+v"(tt[i & 7]));
            } else if constexpr (op.kind == OP_FMA2) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(tt[i & 7]) : "v"(cv[Q].z), "v"(cv[Q].w));
            } else if constexpr (op.kind == OP_ST) {
                store_row<8 * (i >> 2) + (i & 3)>(ta[i & 3], tt[i & 7], voff0, ldy4, rsrcYe, soffY);
            }
        };

        // ---- one body: half a column block (ring slot 2 PAR + H), see "the schedule of one column block" --------------------------------
        auto body = [&](auto PARC, auto HC, auto XRC, int blk) {
            constexpr int PAR = decltype(PARC)::value, H = decltype(HC)::value, Q = PAR ^ 1;
            constexpr int XR = decltype(XRC)::value;                 // >= 0: the pieces carry round XR of the next strip's rows, not the W stream
            constexpr int RSLOT = 2 * PAR + H;                                   // this body's ring slot
            constexpr int NSLOT_R = (RSLOT + 1) & 3;                             // the next body's
            constexpr int DSLOT = (RSLOT + 3) & 3;                               // the slot refilled here: the previous body's
            const unsigned soff_here = std::min(soffW, img_bytes - SLOT);
            soffW += SLOT;
            if constexpr (H == 0) soffY = (unsigned)(blk - 1) * 128u;
            static_for<0, 16>([&](auto JC) {
                constexpr int J = decltype(JC)::value;
                constexpr int G = 16 * H + J;                                    // sub-step of the block: strip registers 8 G ..
                if constexpr (J == BAR_J) {
                    // (the segment's last body has no next slot to wait for: what is in flight is the next strip's)
                    if constexpr (XR != 2) wait_vm<PLAN.vm_bar[H]>();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                wait_lgkm<2 * (FD - 1)>();                                       // the two fragments of this sub-step have arrived
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 3>([&](auto MC) {
                    constexpr int M = decltype(MC)::value, SG = 48 * H + 3 * J + M;
                    // lo.hi, hi.lo (small terms first), hi.hi
                    if constexpr (M == 0) mfma_strip<8 * G + 4, (G == 0)>(acc[PAR], fr[G % NSETS][0]);
                    else if constexpr (M == 1) mfma_strip<8 * G, false>(acc[PAR], fr[G % NSETS][1]);
                    else mfma_strip<8 * G, false>(acc[PAR], fr[G % NSETS][0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (M == 0) {
                        // fragments of the sub-step FD ahead (this slot, or the next one behind the barrier)
                        constexpr int JN = J + FD;
                        constexpr int RS = JN < 16 ? RSLOT : NSLOT_R, JJ = JN & 15, GN = (G + FD) % NSETS;
                        constexpr int OFFS = (RS & 1) * SLOT + JJ * 1024;
                        if constexpr (RS < 2) {
                            lds_read128<OFFS>(fr[GN][0], xa0);
                            lds_read128<OFFS + PLANE>(fr[GN][1], xa0);
                        } else {
                            lds_read128<OFFS>(fr[GN][0], xa1);
                            lds_read128<OFFS + PLANE>(fr[GN][1], xa1);
                        }
                    }
                    if constexpr (slot_cvload(SG)) {
                        // this block's lane constants (used by its epilogue, in the next block's bodies or in the drain)
                        const unsigned so = (unsigned)blk * 512u;
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(cv[PAR]) : "v"(cvoff), "s"(rsrcV), "s"(so) : "memory");
                    }
                    if constexpr (XR == 0 && M == 1 && J < 8) {
                        // round 3 of the next strip, to the staging-only buffer (no ring slot: no barrier to respect)
                        dma_piece<XB_OFF + J * 1024>(xvn[J], rsrcXn, 3u * 256u, ldsw);
                    }
                    constexpr int dp = slot_piece(SG);
                    if constexpr (dp >= 0) {
                        // the W stream's piece dp of the slot three ahead -- or, in the segment's last three bodies, instruction dp of round
                        // XR of the next strip's rows; same LDS target either way
                        if constexpr (XR >= 0) dma_piece<DSLOT * SLOT + dp * 1024>(xvn[dp], rsrcXn, (unsigned)(256 * XR), ldsw);
                        else dma_piece<DSLOT * SLOT + dp * 1024>(lane16 + (unsigned)(dp * 1024), rsrcW, soff_here, ldsw);
                    }
                    static_for<PLAN.begin[SG], PLAN.begin[SG + 1]>([&](auto IC) {
                        __builtin_amdgcn_sched_barrier(0);
                        epi_item(std::integral_constant<int, Q>{}, IC, IM1{});
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        };

        auto drain = [&](auto QC, auto EXTRA, int blk) {
            STAMP(5);
            mfma_drain_nops();
            rsrcYe = rsrcY;
            soffY = (unsigned)blk * 128u;
            static_for<0, STREAM.n>([&](auto IC) { epi_item(QC, IC, EXTRA); });
        };
        // The segment's blocks: a plain loop, then -- when a next segment exists -- its last two blocks as the copies whose last three bodies
        // fetch the next strip's first three rounds (compile-time copies: the plain loop keeps its one branch per two blocks).
        auto plain = [&](auto PARC, int b) {
            body(PARC, I0{}, IM1{}, blk0 + b);
            body(PARC, I1{}, IM1{}, blk0 + b);
        };
        auto tail = [&](auto PARC, int b) {
            constexpr int P = decltype(PARC)::value;
            body(PARC, I0{}, IM1{}, blk0 + b);
            body(PARC, I1{}, I0{}, blk0 + b);
            rsrcYe = rsrcY;
            body(std::integral_constant<int, P ^ 1>{}, I0{}, I1{}, blk0 + b + 1);
            body(std::integral_constant<int, P ^ 1>{}, I1{}, I2{}, blk0 + b + 1);
            // Every wave is done with the ring (the segment's barrier, taken here instead of behind the drain): round 4 goes to the last
            // body's slot and flies while the drain computes.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            static_for<0, 8>([&](auto TC) {
                constexpr int t = decltype(TC)::value;
                dma_piece<(2 * (P ^ 1) + 1) * SLOT + t * 1024>(xvn[t], rsrcXn, 4u * 256u, ldsw);
            });
            drain(std::integral_constant<int, P ^ 1>{}, std::integral_constant<int, 8>{}, blk0 + b + 1);
        };
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[1][e] = 0.0f;
        {
            const int m = has_next ? n - 2 : n;          // blocks of the plain loop
            int b = 0, par = 0;                          // par: parity of the block behind the plain loop
            if (m > 0) {
#pragma nounroll
                for (;;) {
                    plain(I0{}, b);
                    rsrcYe = rsrcY;
                    if (++b >= m) { par = 1; break; }
                    plain(I1{}, b);
                    if (++b >= m) { par = 0; break; }
                }
            }
            if (has_next) {
                if (par == 0) tail(I0{}, b); else tail(I1{}, b);
            } else {
                if (par == 1) drain(I0{}, I0{}, blk0 + b - 1); else drain(I1{}, I0{}, blk0 + b - 1);
            }
        }
        // segment end: nothing of this wave may still be in flight towards the fragment registers.  (Vector memory is NOT drained: the
        // stores need no wait, and the pieces in flight -- the next strip's first rounds -- go to this wave's own parts of the ring, which
        // nobody else touches before the barrier behind the next strip load, whose counted waits cover them.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s8 = 0; s8 < NSETS; ++s8) asm volatile("" ::"v"(fr[s8][0]), "v"(fr[s8][1]));
        if (!has_next) __builtin_amdgcn_s_barrier();   // every wave is done with the ring before the next segment's strip staging refills it
        STAMP(6);
#ifdef LAFF_FCS_TRACE
        ++trc_seg;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA of this workgroup outlives it
    m0_set(m0_keep);
    agpr_hold_end(hold);
#undef STAMP
}

// ---- W [D][512] fp32 (+ bias, BatchNorm, activation) -> the LDS image + the lane-constant table -------------------------------------
// One workgroup (256 threads) per output column: row maximum -> power-of-two scale (as split_rows_kernel), hi / lo halves scattered
// into the image {block n / 32}{K half}{plane}{sub-step}{lane = 32 hh + n % 32}{8 halves}.
__global__ __launch_bounds__(256) void fc_strip_pack_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                            const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int D,
                                                            int act, _Float16* __restrict__ img, float* __restrict__ vec) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    if (n >= D) return;
    const float x0 = W[(size_t)n * ldw + tid], x1 = W[(size_t)n * ldw + 256 + tid];
    float m = fmaxf(fabsf(x0), fabsf(x1));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int e = split_exponent(m);
    const float s = pow2f(9 - e), cs = pow2f(e - 9);
    const int blk = n >> 5, n31 = n & 31;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = t * 256 + tid;
        const float x = (t ? x1 : x0) * s;
        const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
        const int j = k >> 4, w = k & 15;
        const int half = j >> 4, jj = j & 15, hh = (w >> 2) & 1, pos = (w & 3) + 4 * (w >> 3);
        const size_t base = ((size_t)(blk * 2 + half) * 2) * (PLANE / 2);        // in halves: slot start
        const size_t at = (size_t)(jj * 64 + 32 * hh + n31) * 8 + pos;
        img[base + at] = h;
        img[base + PLANE / 2 + at] = l;
    }
    if (tid == 0) {
        const float b = bias ? bias[n] : 0.f, g = bn_scale ? bn_scale[n] : 1.f, t = bn_shift ? bn_shift[n] : 0.f;
        float4 c;
        if (act == LAFF_ACT_TANH) c = make_float4(cs * 2.885390081777927f, b * 2.885390081777927f, -2.f * g, g + t);
        else if (act == LAFF_ACT_SIGMOID) c = make_float4(cs * -1.4426950408889634f, b * -1.4426950408889634f, g, t);
        else if (act == LAFF_ACT_RELU) c = make_float4(cs, b, g, t);
        else c = make_float4(cs * g, fmaf(b, g, t), 0.f, 0.f);
        *(float4*)(vec + 4 * (size_t)n) = c;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
int g_fc_strip = 1;          // LAFF_FC_STRIP (read when a ctx is created; the host's 0 = never take the strip form)

size_t fc_strip_image_bytes(int D) { return (size_t)(D / 32) * (2 * SLOT) + (size_t)D * 16; }
size_t fc_strip_vec_offset(int D) { return (size_t)(D / 32) * (2 * SLOT); }

hipError_t launch_fc_strip_pack(const float* W, int ldw, const float* bias, const float* bn_scale, const float* bn_shift, int D, int act,
                                void* img, hipStream_t st) {
    hipLaunchKernelGGL(fc_strip_pack_kernel, dim3((unsigned)D), dim3(256), 0, st, W, ldw, bias, bn_scale, bn_shift, D, act,
                       (_Float16*)img, (float*)((char*)img + fc_strip_vec_offset(D)));
    return hipGetLastError();
}

hipError_t launch_fc_strip(FcStripArgs& a, int act, hipStream_t st) {
    long U = 0;
    for (int i = 0; i < a.count; ++i) {
        a.p[i].unit0 = (int)U;
        U += (long)((a.p[i].N + FR - 1) / FR) * a.nblk;
    }
    if (U == 0) return hipSuccess;
    if (U >= (1l << 31)) return hipErrorInvalidValue;
    a.total_units = (int)U;
    a.trace = nullptr;
#ifdef LAFF_FCS_TRACE
    if (const char* e = getenv("LAFF_GEMM_TRACE_PTR")) a.trace = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
    const int G = (int)std::min<long>(std::min(g_num_cus, STRIP_MAX_WG), U);
    a.nranges = G;
    // Workgroup b runs on XCD b % 8 (observed; only speed depends on it): XCD x takes the contiguous eighth x of the ranges, so its L2
    // holds the W images of the one or two features its CUs are working on.
    for (int b = 0; b < G; ++b) a.range_of_wg[b] = (unsigned short)((G % 8 == 0) ? (b % 8) * (G / 8) + b / 8 : b);
#define LAFF_FCS_LAUNCH(K)                                                                                                    \
    do {                                                                                                                      \
        static unsigned long long attr_done = 0;                                                                              \
        if (hipError_t e = smem_attr_once(attr_done, fc_strip_kernel<K>, SMEM); e != hipSuccess) return e;                    \
        hipLaunchKernelGGL((fc_strip_kernel<K>), dim3((unsigned)G), dim3(256), SMEM, st, a);                                  \
    } while (0)
    if (act == LAFF_ACT_TANH || act == LAFF_ACT_SIGMOID) LAFF_FCS_LAUNCH(ACT_EXP);
    else if (act == LAFF_ACT_RELU) LAFF_FCS_LAUNCH(ACT_RELU);
    else LAFF_FCS_LAUNCH(ACT_LIN);
#undef LAFF_FCS_LAUNCH
    return hipGetLastError();
}

}  // namespace laff
