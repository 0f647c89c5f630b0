// sim_strip.hip -- the similarity GEMM at K = 512 (one head of d = 512: BASELINE's headline shape) as a STRIP kernel for gfx950.
//
//   S[t][v] = scale * sum_k T[t][k] * V[v][k]       (loss.cosine_sim + get_txt2vis_matrix, /root/reference/loss.py:30-34,
//                                                     /root/reference/model/model.py:1003-1016; fused exact-rank count in the form
//                                                     of /root/reference/predictor.py:232-244)
//
// The tiled kernel (gemm_nt.hip) spends ~45k cycles per 256 x 256 tile for 16.4k cycles of MFMA issue at K = 512: per-tile prologue,
// first K-step, barrier skew and an epilogue with nothing running under it (one workgroup per CU).  Here the decomposition is different:
//   * a workgroup is 4 wavefronts, ONE PER SIMD (512 registers each).  A wavefront keeps 64 text rows x the WHOLE K = 512 in
//     registers as MFMA B fragments (256 registers, the AGPR half) for as long as it works on that strip of 256 rows;
//   * only the video operand moves: column blocks of 32 videos (32 KiB) stream through a three-slot LDS ring by LDS-DMA, continuously
//     across column blocks -- no per-tile prologue, half the LDS-DMA bytes per flop, 0.5 LDS fragment reads per MFMA (one read feeds
//     the two MFMAs of 32x32x16 of a sub-step);
//   * two accumulator sets (2 x 32 registers) alternate: while the MFMAs of column block b fill one set, the epilogue of block b-1
//     (scale, band counters, slab transpose, 64-byte row stores) is issued from the same wavefront in the shadow of those MFMAs, a
//     few instructions behind each one -- a lone wavefront per SIMD has nothing else to cover them.  Where each epilogue
//     instruction goes is decided at compile time (make_epi_plan) together with the operands of the hand-counted s_waitcnt;
//   * the persistent grid (one workgroup per CU) cuts the (strip, column block) sequence into equal contiguous ranges, so every CU
//     gets the same number of column blocks whatever the shape; ranges are handed to workgroups such that the CUs of one XCD walk
//     the video operand at nearly the same column phase (their L2 sees each video row once).
//
// Exact-rank ("banded") count, same contract as gemm_nt.hip's epilogue_banded (rank.hip has the other two launches): per row,
// count[row] += #{col : x > hi}; groups of 16 accumulators that may hold a value inside [lo, hi] are DUMPED (raw values + thresholds)
// for laff_rank_resolve, which applies the precise band test and re-scores exactly.  The per-element work in this kernel is
//   t = hi - x ; sign bit of t -> shift register (v_alignbit) ; unsigned min of the t bits (v_min3_u32: in band <=> 0 <= t <= w)
// i.e. 2.5 vector instructions per accumulator instead of 4, and no per-element branch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

#include "kernels.h"

#ifndef LAFF_STRIP_ABL
#define LAFF_STRIP_ABL 0       // ablation builds (timing only, wrong results): 1 = no fragment reads, 2 = no refill DMA / block barrier
#endif

namespace laff {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) u32x4 gu32x4;
typedef __attribute__((address_space(1))) float gf32;
typedef __attribute__((address_space(1))) f32x4 gf32x4;

constexpr int SR = STRIP_ROWS;                 // text rows per strip: 4 wavefronts x 64
constexpr int CB = STRIP_COLS;                 // videos per column block
constexpr int KBYTES = 1024;                   // K = 512 16-bit elements
constexpr int STAGE = CB * KBYTES;             // one ring slot: 32 KiB
constexpr int RING = 3;                        // ring slots: block b lives in slot b % 3
constexpr int PIECES = CB / 4;                 // 1 KiB LDS-DMA pieces (= columns) per wave per block
constexpr int SLAB_OFF = RING * STAGE;         // per-wave fp32 transpose slabs, one per job (32 rows x 32 columns) of a column block:
constexpr int SLAB_JOB = 32 * 128;             // 128-byte rows, 16-byte chunk c of row r at position c ^ (r & 7) -- conflict-free for the
constexpr int SLAB_BYTES = 2 * SLAB_JOB;       // quad-column ds_write_b128 (8 rows at a time, banks = dword mod 32) and the row-wise ds_read_b128
constexpr int BAND_OFF = SLAB_OFF + 4 * SLAB_BYTES;       // band maximum of every aligned group of 64 columns (fp32)
constexpr int DUMP_OFF = BAND_OFF + STRIP_MAX_GROUPS * 4;  // per-wave staging of one chunk of dumped groups (STRIP_CHUNK entries)
constexpr int DUMP_BYTES = STRIP_CHUNK * STRIP_ENTRY_WORDS * 4;
constexpr int SMEM = DUMP_OFF + 4 * DUMP_BYTES;
static_assert(SMEM <= 160 * 1024 && CB == 32 && DUMP_BYTES % 1024 == 0, "LDS budget / block width / whole-wave flush passes");

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

template <int IMM>
__device__ __forceinline__ void lds_read128(u32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int IMM>
__device__ __forceinline__ void lds_write128(unsigned addr, const f32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void lds_read128_a(u32x4& d, unsigned addr) {        // ... into the accumulator half of the register file
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d) : "v"(addr), "n"(IMM));
}
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one 1 KiB piece of the video stream: the 1,024 bytes (whole K) of column P of this wave's 8 columns of a block, straight into its
// LDS row.  LDS-DMA writes lane L at (M0 base) + 16 L, so the bank swizzle is applied to the SOURCE: LDS slot L of column c holds the
// row's 16-byte chunk L ^ (c & 15) (the fragment reads undo it; c & 15 = 8 (wave & 1) + P, the wave part is folded into lane16x).
// Buffer form: the descriptor starts at the wave's first column of the block and ends with the operand, so columns beyond the last
// video read as zeros (the range check looks at the VGPR + instruction offset only, which is why the block and wave offsets live in
// the descriptor and not in an SGPR offset).
template <int P>
__device__ __forceinline__ void dma_piece(unsigned lane16x, u32x4 rsrc, unsigned m0base) {
    asm volatile("" : "+v"(lane16x));             // opaque: one v_xor per piece instead of 8 hoisted offsets held in registers
    const unsigned voff = lane16x ^ (unsigned)((P << 4) | (P << 10));       // P * 1024 + 16 * (lane ^ column): lane16x < 1024
    asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(m0base), "n"(P * 1024)
                 : "memory", "scc");          // s_add writes SCC: undeclared, hipcc put one of these between an s_add_u32 / s_addc_u32 pair
}
// one 1 KiB piece of the strip staging: lane L's 16 bytes at descriptor offset voff + soff -> LDS (m0base + LDSOFF) + 16 L
template <int LDSOFF>
__device__ __forceinline__ void stage_piece(unsigned voff, u32x4 rsrc, unsigned soff, unsigned m0base) {
    asm volatile("s_add_i32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "s"(m0base), "n"(LDSOFF) : "memory", "scc");
}
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

// keep a wave-uniform value in an SGPR and opaque (no re-load from the kernarg segment inside the hand-counted loops)
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

// MFMA from inline asm with the register FILES pinned: the strip fragment (B operand) must live in the accumulator half of the
// register file ("a": 256 of them are the whole point of this kernel) and the accumulators in the architectural half ("v": the
// epilogue's vector instructions read them directly).  Left to itself hipcc does the opposite -- accumulators to AGPRs, the 256
// strip registers to VGPRs, 160 of them spilled.  hipcc knows nothing about an asm MFMA's latency: every consumer of `acc` sits
// behind other MFMAs (>= 2 of them: 64 cycles) or behind mfma_drain_nops().
template <int MODE, bool FIRST>
__device__ __forceinline__ void mfma16(f32x16& acc, const u32x4& colfrag, const u32x4& stripfrag) {
    if constexpr (MODE == GEMM_F16) {
        if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(colfrag), "a"(stripfrag));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(colfrag), "a"(stripfrag));
    } else {
        if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(colfrag), "a"(stripfrag));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(colfrag), "a"(stripfrag));
    }
}
__device__ __forceinline__ void mfma_drain_nops() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// ---- the schedule of one column block, all compile time ---------------------------------------------------------------------------
// A block is 32 sub-steps J (one 16-element K slice each) of 2 MFMAs (M = row block rb).  Slot sigma = 2 J + M is the issue shadow of
// MFMA M of sub-step J; what is placed in a slot is issued right behind that MFMA.
//
// Pinned fillers:
//   * fragment reads, FDIST sub-steps ahead into 8 register sets: slot (J, 0) asks for sub-step J + FDIST of this block (J <= 31 -
//     FDIST); the first FDIST sub-steps of the NEXT block are asked for in sub-steps BAR_J .. BAR_J + 2, one per slot, behind the
//     block barrier;
//   * the band value of this block's 64-column group (one ds_read_b32) right behind the block's last fragment read;
//   * the block barrier in front of sub-step BAR_J: counted lgkmcnt (this block's fragments have all arrived), counted vmcnt (the next
//     block's pieces have landed; the epilogue's stores issued since may stay in flight), s_barrier.  It frees this block's ring
//     slot -- the two late DMA pieces (block b + 3) go to sub-step 31, the other six to sub-steps 1..6 of the next body;
//   * the thresholds of this block (computed behind the barrier, read by its epilogue in the NEXT body, whose stream therefore
//     never reaches sub-step BAR_J).
constexpr int NSLOT = 64;
constexpr int FDIST = 6;
constexpr int BAR_J = 28;
constexpr int slot_read_sub(int sg) {        // sub-step whose fragment is asked for in this slot: 0..31 this block, 32 + j next block, -1 none
    const int J = sg >> 1, M = sg & 1;
    if (M == 0 && J + FDIST < 32) return J + FDIST;
    if (J >= BAR_J && J < BAR_J + 3) return 32 + 2 * (J - BAR_J) + M;
    return -1;
}
constexpr int slot_dma_piece(int sg) {       // DMA piece issued in this slot: 0, 1 late (block b + 3); 2..7 early (block b + 2); -1 none
    const int J = sg >> 1, M = sg & 1;
    if (J == 31) return M;
    if (M == 1 && J >= 1 && J <= 6) return J + 1;
    return -1;
}
constexpr int BAND_READ_SLOT = 2 * (31 - FDIST);       // the slot of this block's last fragment read

// The epilogue of a column block as a static stream of micro-ops.  Job = row block rb (32 x 32 accumulators).  Per accumulator e:
//   SUBALN  t = hi - x ; sh = (sh << 1) | sign(t) ; MIN (odd e)  m = umin(m, bits(t_prev), bits(t))            (banded count)
//   DSW  after every 4th: the quad times scale, ds_write_b128 into the job's slab                                 (scores wanted)
// then CHK (any 0 <= t <= w in the job?  -> dump, rare), CNT (count += popcount of the 16 sign bits), DSR i (slab rows 8 i .. 8 i + 7
// back, 128 bytes per row, into 4 registers of its own), and per read WAITR + STG (non-temporal 16-byte buffer store, 8 rows x 128
// bytes per instruction).  Job 0 is stored from job 1's element stream and job 1 at the end, so that the LDS latency of the
// read-back is covered and the stores are spread over the block.
enum : unsigned char { OP_MUL = 0, OP_SUBALN, OP_MIN, OP_DSW, OP_CHK, OP_CNT, OP_DSR, OP_WAITR, OP_STG, OP_GAP };
constexpr int op_cost(unsigned char k) { return k == OP_DSW ? 5 : (k == OP_SUBALN ? 2 : 1); }      // (DSW: 4 products + the LDS store unless scale == 1)       // instructions of one stream item
struct EpiOp { unsigned char kind, job, arg; };
constexpr int EPI_MAX_OPS = 256;
constexpr int TAIL_GAP = 3;
struct EpiStream { EpiOp op[EPI_MAX_OPS]; int n; };

template <bool HAVE_S, bool BANDED>
constexpr EpiStream make_epi_stream() {
    EpiStream s{};
    s.n = 0;
    auto push = [&](unsigned char k, int job, int arg) { s.op[s.n++] = EpiOp{k, (unsigned char)job, (unsigned char)arg}; };
    for (int job = 0; job < 2; ++job) {
        for (int e = 0; e < 16; ++e) {
            if (BANDED) {
                push(OP_SUBALN, job, e);
                if (e >= 3 && (e & 1)) push(OP_MIN, job, e - 2);         // the pair (e - 3, e - 2): two statements behind its second SUBALN
            }
            if (HAVE_S) {
                if ((e & 3) == 3) push(OP_DSW, job, e >> 2);
                // Job 0's four stores ride in job 1's element stream, one every fourth element: a store instruction is 8 rows x 128
                // bytes, whole cache lines where the row pitch allows (as 16 rows x 64 bytes the same bytes left the chip at 3.2 TB/s,
                // this way at 4.0: tools/probe/stprobe), and a row of the slab is complete only behind the job's last quad.
                if (job > 0 && (e & 3) == 3) { push(OP_WAITR, 0, e >> 2); push(OP_STG, 0, e >> 2); }
            }
        }
        if (BANDED) { push(OP_MIN, job, 15); if (job == 1) push(OP_CHK, 1, 0); push(OP_CNT, job, 0); }
        if (HAVE_S) for (int i = 0; i < 4; ++i) push(OP_DSR, job, i);
    }
    if (HAVE_S)
        for (int i = 0; i < 4; ++i) {
            // job 1's stores: the tail of the stream, TAIL_GAP slots apart
            if (i) for (int g = 0; g < TAIL_GAP; ++g) push(OP_GAP, 1, 0);
            push(OP_WAITR, 1, i); push(OP_STG, 1, i);
        }
    return s;
}

struct EpiPlan {
    short begin[NSLOT + 1];    // stream range of slot sigma: [begin[sigma], begin[sigma + 1])
    short wait_frag[32];       // lgkmcnt operand in front of sub-step J: LDS operations issued after the fragment read of J
    short wait_r[2][4];        // ... in front of the store of (job, read i): LDS operations issued after that slab read
    short lgkm_bar, vm_bar;    // operands of the block barrier's waits
    bool fits;
};

// Places the stream (EPI) and counts, for a body that follows an identical body, how many LDS / VMEM operations are issued between
// an operation and the wait that needs it (LDS operations of a wavefront complete in order; so do its vector-memory operations).
template <bool HAVE_S, bool BANDED, bool EPI>
constexpr EpiPlan make_epi_plan() {
    EpiPlan p{};
    EpiStream st{};
    st.n = 0;
    if (EPI) st = make_epi_stream<HAVE_S, BANDED>();
    // uniform density over the usable slots (a lone wavefront hides only a handful of instructions behind each MFMA: a stream
    // packed into the first half of the block left 8 instructions per slot there and nothing behind)
    int usable = 0;
    for (int sg = 0; sg < NSLOT; ++sg)
        if ((sg >> 1) < BAR_J && slot_dma_piece(sg) < 0) ++usable;
    int total_cost = 0;
    for (int i = 0; i < st.n; ++i) total_cost += op_cost(st.op[i].kind);
    int at = 0, k = 0, spent = 0;
    for (int sg = 0; sg < NSLOT; ++sg) {
        p.begin[sg] = (short)at;
        if ((sg >> 1) >= BAR_J || slot_dma_piece(sg) >= 0) continue;   // the thresholds change at the barrier; DMA slots are full
        ++k;
        const int spread = usable - (HAVE_S ? 3 * TAIL_GAP + 4 : 2);  // the gap items of the tail each cost a slot
        const int target = (int)(((long)k * total_cost + spread - 1) / spread);
        while (at < st.n && spent < target) {
            spent += op_cost(st.op[at].kind);
            if (st.op[at].kind == OP_GAP) { ++at; break; }       // a gap item closes the slot
            ++at;
        }
    }
    p.begin[NSLOT] = (short)at;
    p.fits = at == st.n;
    // program order over TWO consecutive bodies (the second one is the body being planned)
    int lds_n = 0, vm_n = 0;
    int frag_ord[2][40] = {};           // [body][sub-step index as returned by slot_read_sub]: LDS ordinal of that read
    int band_ord[2] = {0, 0};
    int lds_before[2 * NSLOT + 1] = {}, vm_before[2 * NSLOT + 1] = {};
    int dsr_ord[2][4] = {}, waitr_at[2][4] = {};
    int last_early_vm[2] = {0, 0};
    for (int body = 0; body < 2; ++body)
        for (int sg = 0; sg < NSLOT; ++sg) {
            lds_before[body * NSLOT + sg] = lds_n;
            vm_before[body * NSLOT + sg] = vm_n;
            const int rs = slot_read_sub(sg);
            if (rs >= 0) frag_ord[body][rs] = ++lds_n;
            if (BANDED && sg == BAND_READ_SLOT) band_ord[body] = ++lds_n;
            const int dp = slot_dma_piece(sg);
            if (dp >= 0) {
                ++vm_n;
                if (dp == PIECES - 1) last_early_vm[body] = vm_n;
            }
            for (int i = p.begin[sg]; i < p.begin[sg + 1]; ++i) {
                const EpiOp o = st.op[i];
                if (o.kind == OP_DSW || o.kind == OP_DSR) ++lds_n;
                if (o.kind == OP_STG) ++vm_n;
                if (body == 1) {
                    if (o.kind == OP_DSR) dsr_ord[o.job][o.arg] = lds_n;
                    if (o.kind == OP_WAITR) waitr_at[o.job][o.arg] = lds_n;
                }
            }
        }
    auto clamp = [](int c, int hi) { return c < 0 ? 0 : (c > hi ? hi : c); };
    for (int J = 0; J < 32; ++J) {
        // one wait per PAIR of sub-steps, in front of the even one, for the odd one's fragment (asked for later: it covers both).
        // That fragment: asked for in this body (>= FDIST) or as "next block" sub-step 32 + j in the previous one
        const int Jn = J | 1;
        const int ord = Jn >= FDIST ? frag_ord[1][Jn] : frag_ord[0][32 + Jn];
        p.wait_frag[J] = (short)clamp(lds_before[NSLOT + 2 * (J & ~1)] - ord, 15);
    }
    {
        // the barrier: this block's last fragment read (and the band value right behind it) ...
        const int ord = BANDED ? band_ord[1] : frag_ord[1][31];
        p.lgkm_bar = (short)clamp(lds_before[NSLOT + 2 * BAR_J] - ord, 15);
        // ... and block b + 1 complete: its last piece was issued as the last early piece of the PREVIOUS body
        p.vm_bar = (short)clamp(vm_before[NSLOT + 2 * BAR_J] - last_early_vm[0], 63);
    }
    for (int k = 0; k < 2; ++k)
        for (int h = 0; h < 4; ++h) p.wait_r[k][h] = (short)clamp(waitr_at[k][h] - dsr_ord[k][h], 15);
    return p;
}

}  // namespace

// MODE: GEMM_F16 / GEMM_BF16.  BANDED: exact-rank count + dumps.  HAVE_S: the fp32 score matrix is written (SCALE1: scale == 1).
// Debug builds (timing only unless noted; tools/debug/build_strip_variant.sh): -DLAFF_STRIP_SERIAL = the K loops alone, no epilogue;
// -DLAFF_STRIP_ABL = 1 no fragment reads / 2 no refill DMA and barrier / 3 neither; -DLAFF_STRIP_NOCHK = no band test, -DLAFF_STRIP_NODUMPBODY
// = band test without the dump, -DLAFF_STRIP_NOSTG = no score stores; -DLAFF_STRIP_TRACE = cycle stamps (correct results);
// -DLAFF_STRIP_STFLAVOR="..." = cache policy bits of the score stores (correct results).
template <int MODE, bool BANDED, bool HAVE_S, bool SCALE1>
__global__ __launch_bounds__(256, 1) void sim_strip_kernel(const StripArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);

#ifdef LAFF_STRIP_TRACE
    // debug build: cycle stamps of wave 0 -- [0] start, [1] table staged; per segment s (up to 3): base 2 + 20 s: +0 start, +1 strip loads
    // issued, +2 prologue barrier passed, +3 + k after the K loop of block k (k < 12), +16 segment end, +17 n
    unsigned long long* const trc = a.trace ? a.trace + (size_t)blockIdx.x * 64 : nullptr;
    int trc_seg = 0;
#define STAMP(i) do { if (trc && tid == 0 && (i) < 64) trc[(i)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
    STAMP(0);
    // ---- arguments, pinned in SGPRs (nothing may be re-loaded from the kernarg segment inside the hand-counted loops: scalar loads
    // share lgkmcnt with the LDS reads and return out of order) -------------------------------------------------------------------
    const int nR = pin_s(a.nR), nC = pin_s(a.nC), ldo = pin_s(a.ldo), col0 = pin_s(a.col0);
    const float scale = pin_s(a.scale);
    const float inv_scale = pin_s(a.inv_scale);
    const unsigned long long pT = pin_s((unsigned long long)a.T), pV = pin_s((unsigned long long)a.V);
    const unsigned long long pOut = pin_s((unsigned long long)a.out), pPairs = pin_s((unsigned long long)a.pairs);
    const int NB = (nC + CB - 1) / CB, NS = (nR + SR - 1) / SR;
    const long U = (long)NS * NB;
    const int nranges = pin_s(a.nranges);
    const int myrange = a.range_of_wg[blockIdx.x];
    long u0 = U * myrange / nranges;
    const long u1 = U * (myrange + 1) / nranges;

    // the dump list: header {chunks taken from the pool, overflow flag, NW | 1 << 31, NCH} | NCH per-chunk entry counts | NCH chunks of
    // STRIP_CHUNK entries of STRIP_ENTRY_WORDS words.  Wavefront w starts in chunk w; a full chunk is closed (its count written) and
    // the next one taken from the pool with ONE atomic per STRIP_CHUNK entries (the atomic's return drains this wave's memory queue:
    // per entry, as a first version had it behind small fixed segments, it cost a third of the launch).
    const unsigned NW = (unsigned)nranges * 4u;
    const unsigned total_words = 2u * pin_s(a.pair_cap);
    // chunks that fit behind the header and the count table (rounded up to 4 words): NCH + 3 + NCH * chunk words <= total_words
    const unsigned NCH = pin_s(total_words >= 3u ? (total_words - 3u) / (1u + STRIP_CHUNK * STRIP_ENTRY_WORDS) : 0u);
    const unsigned cnt_words = (NCH + 3u) & ~3u;
    const unsigned wave_global = (unsigned)blockIdx.x * 4u + (unsigned)wave;
    unsigned cur_chunk = wave_global, cur_n = 0;                            // wave-uniform: the chunk being filled, entries in it
    const unsigned list_base = pin_s((4u + cnt_words) * 4u);                // byte offset of chunk 0 in the list
    u32x4 rsrcP = {0, 0, 0, 0x00020000u};                                   // the whole list as a raw buffer
    if constexpr (BANDED) {
        rsrcP = rebased_rsrc(pPairs, 16ull + 8ull * a.pair_cap, 0ull);
        if (blockIdx.x == 0 && tid == 0) {
            gu32* pp = (gu32*)pPairs;
            pp[2] = NW | 0x80000000u;
            pp[3] = NCH;
        }
        // band maxima of the aligned 64-column groups -> LDS (laff_rank_prepare stores them behind the per-column values)
        const float* bm = a.band_c + ((nC + 3) & ~3);
        float* dst = (float*)(smem + BAND_OFF);
        for (int i = tid; i < (nC + 63) / 64; i += 256) dst[i] = bm[i];
    }
    // a staged chunk of dumped groups -> the list (see dump_group)
    const unsigned dump_base = lds0 + DUMP_OFF + (unsigned)wave * DUMP_BYTES;
    auto flush_chunk = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the staged entries are written
        const unsigned cbase = cur_chunk < NCH ? list_base + cur_chunk * (unsigned)DUMP_BYTES : 0x80000000u;     // (no chunk: dropped by the bounds check)
        unsigned lane16 = (unsigned)lane * 16u;
        asm volatile("" : "+v"(lane16));
#pragma unroll 1
        for (unsigned k = 0; k < (unsigned)DUMP_BYTES; k += 1024u) {
            u32x4 t;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(dump_base + lane16 + k) : "memory");
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(t), "v"(cbase + lane16 + k), "s"(rsrcP) : "memory");
        }
    };
    __syncthreads();
    STAMP(1);
    const unsigned long long vbytes = (unsigned long long)(unsigned)nC * (unsigned)KBYTES;     // the video operand: 1,024-byte rows

    // fragment-read addresses: column c31 of the block, logical 16-byte chunk 2 j + hh of sub-step j = 8 a + bb lives in slot
    // (16 a) + ((2 bb + hh) ^ (c31 & 15)): 8 per-lane addresses (of the ring slot in use), a rides in the instruction's offset field
    unsigned X[8];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) X[bb] = lds0 + (unsigned)l31 * KBYTES + (unsigned)(((2 * bb + hh) ^ (l31 & 15)) * 16);
    const unsigned lane16x = ((unsigned)lane * 16u) ^ ((unsigned)(wave & 1) << 7);       // 16 * (lane ^ 8 (wave & 1))
    const unsigned m0_keep = m0_get();
    // slab addresses of this wave: written in accumulator layout (row l31, 4 consecutive columns per quad), read back as rows
    const unsigned slab0 = lds0 + SLAB_OFF + (unsigned)wave * SLAB_BYTES;
    // lane (l31, hh) holds of text row l31 the column quads 2 q + hh (q = e >> 2): chunk position (2 q + hh) ^ k, k = l31 & 7 -- the 8
    // rows a ds_write_b128 stores in one LDS cycle (lanes 8 g .. 8 g + 7; its banks are dword mod 32 = one 128-byte row) then take 8
    // different positions.  (With k = (l31 >> 1) & 3, chosen for 256-byte banking, every such store was a 2-way conflict: 12.6M of 61M
    // LDS cycles per launch at C4, SQ_LDS_BANK_CONFLICT.)
    const unsigned slab_k = (unsigned)l31 & 7u;
    unsigned slab_w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) slab_w[q] = slab0 + (unsigned)l31 * 128u + ((((unsigned)(2 * q + hh)) ^ slab_k) << 4);
    // read-back i: rows 8 i .. 8 i + 7, a whole 128-byte row per 8 lanes (row & 7 = lane >> 3); the 16 lanes a ds_read_b128 serves per
    // cycle ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md) take half rows of four rows, two per 128-byte half of the banks: no conflict
    const unsigned slab_r = slab0 + (unsigned)(lane >> 3) * 128u + (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned wrow = (unsigned)(PIECES * wave) * KBYTES;              // this wave's 8 columns of a block (LDS and source offset)

#ifndef LAFF_STRIP_SERIAL
    constexpr bool EPI = true;
#else
    constexpr bool EPI = false;
#endif
    constexpr EpiPlan PLAN = make_epi_plan<HAVE_S, BANDED, EPI>();
    constexpr EpiStream STREAM = make_epi_stream<HAVE_S, BANDED>();
    static_assert(PLAN.fits, "the epilogue stream does not fit behind the MFMAs of one column block");

    while (u0 < u1) {
        // declared per segment: nothing of them is carried from one segment to the next (no loop-carried copies)
        u32x4 B[2][32];          // the strip: rows (rb * 32 + l31) of this wave's 64, k = 16 j + 8 hh .. + 7
        f32x16 acc[2][2];        // [set][rb]
        u32x4 fr[8];             // column fragments, 8 sub-steps deep: [j & 7]
        const int strip = (int)(u0 / NB);
        const int cb0 = (int)(u0 - (long)strip * NB);
        const int n = (int)std::min<long>(NB - cb0, u1 - u0);              // column blocks of this segment
        u0 += n;
        const int row0 = strip * SR;
        const int row_w = row0 + wave * 64;                                 // first row of this wave
#ifdef LAFF_STRIP_TRACE
        const int tb = 2 + 20 * trc_seg;
        ++trc_seg;
        if (trc && tid == 0 && tb + 17 < 64) trc[tb + 17] = (unsigned long long)n;
#endif
        STAMP(tb);

        // per-row inputs of the banded count
        float sgf[2] = {0, 0}, brow[2] = {0, 0};
        int gtc[2] = {-1, -1};
        int cnt[2] = {0, 0};
        if constexpr (BANDED) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int r = row_w + rb * 32 + l31;
                const int rc = std::min(r, nR - 1);
                sgf[rb] = (float)a.s_gt64[rc];
                brow[rb] = a.band_r[rc];
                gtc[rb] = r < nR ? a.gt_col[rc] - col0 : -1;
            }
        }
        // ---- the strip -> registers (once per segment): eight K-eighth rounds (64 rows x 128 bytes: one line of every row) by LDS-DMA
        // through three 8 KiB buffers of this wave in the idle ring, then 8 ds_read_b128 per round into the accumulator file.  (As 64
        // direct 16-byte loads per lane at 1 KiB stride -- the fragment layout -- every load instruction touched 32 lines for 32 bytes
        // each: bound by the CU's address path, ~21k cycles per segment.)  Instruction t of a round: rows 8 t + (lane >> 3), LDS chunk
        // lane & 7 of the row <- its logical chunk (lane & 7) ^ (row & 7) (the swizzle on the source side, LDS-DMA writes lane-linear).
        {
            const int rows_w = std::min(64, nR - row_w);                         // rows of this wave inside the matrix (may be <= 0)
            // rows beyond the matrix read as zeros (the range check of this descriptor covers VGPR + scalar offset: measured -- with
            // 896 bytes held back for the scalar round offset the last row lost rounds 1..7)
            const u32x4 rsrcT = rebased_rsrc(pT + (unsigned long long)(rows_w > 0 ? row_w : 0) * KBYTES,
                                             rows_w > 0 ? (unsigned long long)rows_w * KBYTES : 0ull, 0ull);
            unsigned ln = (unsigned)lane;
            asm volatile("" : "+v"(ln));
            const unsigned sv0 = (ln >> 3) * (unsigned)KBYTES + (((ln & 7u) ^ ((ln >> 3) & 7u)) << 4);
            const unsigned stg = pin_s(lds0 + (unsigned)wave * (3u * 8192u));
            unsigned xr[4];                                                       // read-back: row l31 (+ 32 rb), chunk (2 q + hh) ^ (row & 7)
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[q] = stg + (ln & 31u) * 128u + ((((unsigned)(2 * q) + (ln >> 5)) ^ (ln & 7u)) << 4);
            auto issue_round = [&](auto EC) {
                constexpr int e = decltype(EC)::value, buf = e % 3;
                static_for<0, 8>([&](auto TC) {
                    constexpr int t = decltype(TC)::value;
                    stage_piece<buf * 8192 + t * 1024>(sv0 + (unsigned)(t * 8 * KBYTES), rsrcT, (unsigned)(e * 128), stg);
                });
            };
            auto read_round = [&](auto EC) {
                constexpr int e = decltype(EC)::value, buf = e % 3;
                static_for<0, 8>([&](auto IC) {
                    constexpr int rb = decltype(IC)::value >> 2, q = decltype(IC)::value & 3;
                    lds_read128_a<buf * 8192 + rb * 4096>(B[rb][4 * e + q], xr[q]);
                });
                wait_lgkm<0>();                                                   // (the buffer is refilled next)
            };
            using E0 = std::integral_constant<int, 0>; using E1 = std::integral_constant<int, 1>; using E2 = std::integral_constant<int, 2>;
            using E3 = std::integral_constant<int, 3>; using E4 = std::integral_constant<int, 4>; using E5 = std::integral_constant<int, 5>;
            using E6 = std::integral_constant<int, 6>; using E7 = std::integral_constant<int, 7>;
            issue_round(E0{}); issue_round(E1{}); issue_round(E2{});
            wait_vm<16>(); read_round(E0{}); issue_round(E3{});
            wait_vm<16>(); read_round(E1{}); issue_round(E4{});
            wait_vm<16>(); read_round(E2{}); issue_round(E5{});
            wait_vm<16>(); read_round(E3{}); issue_round(E6{});
            wait_vm<16>(); read_round(E4{}); issue_round(E7{});
            wait_vm<16>(); read_round(E5{});
            wait_vm<8>(); read_round(E6{});
            wait_vm<0>(); read_round(E7{});
        }
        __builtin_amdgcn_s_barrier();            // every wave has emptied its staging buffers: the ring prologue may fill the slots
        asm volatile("" ::: "memory");
        // thresholds of the block whose epilogue comes next, [rb], in accumulator units: written behind the block barrier of the
        // block's own K loop, read by its epilogue in the next body (whose stream stays in front of ITS barrier) or by the drain.
        // Initial value: nothing counted, nothing listed -- the first body's epilogue runs on an all-zero accumulator set.
        float thr_lo[2] = {__builtin_inff(), __builtin_inff()}, thr_hi[2] = {__builtin_inff(), __builtin_inff()};
        // epilogue state
        unsigned sh[4] = {0, 0, 0, 0}, mmj[2] = {0, 0};
        float tt[4] = {0, 0, 0, 0};
        u32x4 rr[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        u32x4 rsrcSb = {0, 0, 0, 0x00020000u};      // the score rows of the block whose epilogue is running
        // score rows of this strip as a raw buffer (re-based per column block): rows beyond the matrix are dropped by its bounds check
        unsigned long long s_base = 0ull, s_bytes = 0ull;
        unsigned voffs[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // a store instruction covers 8 rows x 128 bytes (whole lines where the row pitch allows): the wave's 8 row groups
        if constexpr (HAVE_S) {
            s_base = pOut + (unsigned long long)row0 * (unsigned)ldo * 4ull;
            s_bytes = ((unsigned long long)(std::min(SR, nR - row0) - 1) * (unsigned)ldo + (unsigned)nC) * 4ull;
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8)
                voffs[k8] = ((unsigned)(wave * 64 + 8 * k8 + (lane >> 3)) * (unsigned)ldo + (unsigned)(lane & 7) * 4u) * 4u;
        }
        // running descriptors of the block loop: the score columns of block b - 1 (sb_*: base of block cb0 - 1 first, bytes left from
        // block cb0 on) and the video rows of block b + 2 (rsrcNext)
        unsigned long long sb_base = s_base + (unsigned long long)cb0 * (CB * 4u) - CB * 4u;
        unsigned sb_left = (unsigned)(s_bytes - (unsigned long long)cb0 * (CB * 4u));
        u32x4 rsrcNext = rebased_rsrc(pV, vbytes, (unsigned long long)(cb0 + 2) * STAGE + wrow);
        STAMP(tb + 1);

        // ---- ring prologue: blocks 0 and 1 whole, pieces 0, 1 of block 2 (the first body brings in the rest) ---------------------
        {
            const u32x4 r0 = rebased_rsrc(pV, vbytes, (unsigned long long)cb0 * STAGE + wrow);
            const u32x4 r1 = rebased_rsrc(pV, vbytes, (unsigned long long)(cb0 + 1) * STAGE + wrow);
            const u32x4 r2 = rebased_rsrc(pV, vbytes, (unsigned long long)(cb0 + 2) * STAGE + wrow);
            static_for<0, PIECES>([&](auto P) { dma_piece<decltype(P)::value>(lane16x, r0, lds0 + wrow); });
            static_for<0, PIECES>([&](auto P) { dma_piece<decltype(P)::value>(lane16x, r1, lds0 + STAGE + wrow); });
            static_for<0, 2>([&](auto P) { dma_piece<decltype(P)::value>(lane16x, r2, lds0 + 2 * STAGE + wrow); });
        }
        wait_vm<0>();
        // hipcc keeps its own scoreboard of the loads it generated (the strip, the per-row inputs) and would put its `s_waitcnt vmcnt(n)`
        // in front of their FIRST USES -- inside the block loop, where they would drain the DMA pieces and score stores on every pass.
        // Using every loaded register here pins those waits to this spot (where everything has landed anyway).
#pragma unroll
        for (int j = 0; j < 32; ++j) asm volatile("" ::"a"(B[0][j]), "a"(B[1][j]));
        asm volatile("" ::"v"(sgf[0]), "v"(sgf[1]), "v"(brow[0]), "v"(brow[1]), "v"(gtc[0]), "v"(gtc[1]));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(tb + 2);
        // fragments of sub-steps 0 .. FDIST - 1 of block 0 (ring slot 0)
        static_for<0, FDIST>([&](auto JJ) {
            constexpr int j = decltype(JJ)::value;
            lds_read128<(j >> 3) * 256>(fr[j & 7], X[j & 7]);
        });

        // ---- one dumped group: 16 raw accumulators of one lane + what laff_rank_resolve needs to test them {row, colbase, lo, hi |
        // mask16, the row's ground-truth column, 0, 0 | x[16]}.  The groups are staged in LDS, a chunk (STRIP_CHUNK entries) per
        // wavefront, and a full chunk leaves with six whole-wave 16-byte stores: as six global stores per dumped lane the
        // entries sat in the wave's in-order memory queue, where the counted vmcnt waits of the block barrier had to wait for them
        // too -- a store round trip per dump, and a third of the jobs dump (0.03 ms of 0.36 at C4).  LDS writes retire in ~100 cycles.
        auto dump_group = [&](bool hit, int row, int colbase, float lo, float hi, unsigned mask16, int gt_col_of_row, const f32x16& x) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m == 0ull) return;
            const unsigned n = (unsigned)__builtin_popcountll(m);
            if (cur_n + n > STRIP_CHUNK) {                                   // wave-uniform, once per STRIP_CHUNK entries at most
                gu32* pp = (gu32*)pPairs;
                flush_chunk();
                if (lane == 0 && cur_chunk < NCH) pp[4 + cur_chunk] = cur_n;  // close the chunk
                unsigned nc = 0;
                if (lane == 0) nc = NW + __hip_atomic_fetch_add(pp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cur_chunk = __builtin_amdgcn_readfirstlane(nc);
                cur_n = 0;
                if (lane == 0 && cur_chunk >= NCH) pp[1] = 1u;                // pool exhausted: flagged, the entries are dropped
            }
            const unsigned slot = cur_n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            cur_n += n;
#ifdef LAFF_STRIP_NODUMPBODY
            if (false) {
#else
            if (hit) {
#endif
                const unsigned at = dump_base + slot * (STRIP_ENTRY_WORDS * 4u);
                const f32x4 q0 = {x[0], x[1], x[2], x[3]}, q1 = {x[4], x[5], x[6], x[7]}, q2 = {x[8], x[9], x[10], x[11]},
                            q3 = {x[12], x[13], x[14], x[15]};
                lds_write128<32>(at, q0);
                lds_write128<48>(at, q1);
                lds_write128<64>(at, q2);
                lds_write128<80>(at, q3);
                u32x4 h, h2;
                h.x = (unsigned)row; h.y = (unsigned)colbase; h.z = __float_as_uint(lo); h.w = __float_as_uint(hi);
                h2.x = mask16; h2.y = (unsigned)gt_col_of_row; h2.z = 0u; h2.w = 0u;
                // (s_nop behind the header writes: hipcc does not know these are stores, the temporaries may be handed to the next instruction)
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\ts_nop 1" ::"v"(at), "v"(h), "v"(h2) : "memory");
            }
        };

        // ---- generic epilogue of the LAST block of a segment (runtime bounds: partial column blocks, ragged strips).  The ring is idle by
        // then, so the accumulators go through it: each wave writes its 64 x 32 tile row-major into its own quarter of slot 0 (16-byte
        // chunk c of row r at chunk c ^ (r & 7)), then lane L walks row L in two groups of 16 columns -- the groups of the dump format
        // -- in a rolled loop: a few dozen registers and one copy of the code instead of two unrolled 32-accumulator epilogues.
        auto drain_to_lds = [&](auto SETC) {
            constexpr int SET = decltype(SETC)::value;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // this wave's own pieces have landed: its quarter is its own
            unsigned lane_o = (unsigned)lane;
            asm volatile("" : "+v"(lane_o));                                        // addresses made here, not hoisted out of the block loop
            const unsigned tile = lds0 + wrow;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned r = (unsigned)rb * 32u + (lane_o & 31u), c = 2u * q + (lane_o >> 5);
                    const f32x16& x = acc[SET][rb];
                    const f32x4 v = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
                    lds_write128<0>(tile + r * 128u + ((c ^ (r & 7u)) << 4), v);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        auto drain_rows = [&](int cb) {
            const int c0 = cb * CB;
            const bool vec_ok = (ldo & 3) == 0 && (pOut & 15ull) == 0ull;
            unsigned lane_o = (unsigned)lane;
            asm volatile("" : "+v"(lane_o));
            const unsigned rowbase = lds0 + wrow + lane_o * 128u;
            // lane L = row L of the wave's 64 = (rb = hh, l31): its own per-row registers of row block hh
            const float lo = thr_lo[hh], hi = thr_hi[hh];
            const int row = row_w + (int)lane_o, gt = gtc[hh];
            const bool rv = row < nR;
            int c_here = 0;
#pragma unroll 1
            for (int h2 = 0; h2 < 2; ++h2) {
                const int colbase = c0 + 4 * h2;
                f32x16 x;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4 t;
                    const unsigned c = 2u * q + (unsigned)h2;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(rowbase + ((c ^ (lane_o & 7u)) << 4)));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    x[4 * q] = __uint_as_float(t.x); x[4 * q + 1] = __uint_as_float(t.y);
                    x[4 * q + 2] = __uint_as_float(t.z); x[4 * q + 3] = __uint_as_float(t.w);
                }
                if constexpr (BANDED) {
                    unsigned mask16 = 0u;
                    bool has_gt = false;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int col = colbase + 8 * (e >> 2) + (e & 3);
                        const bool valid = rv && col < nC && col != gt;
                        has_gt |= rv && col == gt;
                        c_here += (valid && x[e] > hi) ? 1 : 0;
                        const bool inb = valid && __builtin_amdgcn_fmed3f(x[e], lo, hi) == x[e];
                        mask16 |= inb ? (1u << e) : 0u;
                    }
                    dump_group(mask16 != 0u || has_gt, row, colbase, lo, hi, mask16, gt, x);
                }
                if (HAVE_S && rv) {
                    gf32* orow = (gf32*)pOut + (size_t)row * ldo;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = colbase + 8 * q;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = x[4 * q + e] * scale;
                        if (vec_ok && col + 3 < nC) {
                            f32x4 t = {v[0], v[1], v[2], v[3]};
                            __builtin_nontemporal_store(t, (gf32x4*)(orow + col));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col + e < nC) orow[col + e] = v[e];
                        }
                    }
                }
            }
            cnt[0] += hh == 0 ? c_here : 0;
            cnt[1] += hh == 1 ? c_here : 0;
        };

        // ---- one micro-op of the interleaved epilogue: block `cbp` (accumulator set Q), see make_epi_stream ------------------------
        auto epi_item = [&](auto QC, auto IC, int cbp) {
            constexpr int Q = decltype(QC)::value;
            constexpr EpiOp op = STREAM.op[decltype(IC)::value];
            constexpr int rb = op.job, arg = op.arg;
            // The per-element operations are inline asm: as plain expressions they are pure values to hipcc, which sank them all next to
            // their uses (two slots of 66 and 109 instructions per block, the rest empty) whatever sched_barrier said.
            if constexpr (op.kind == OP_MUL) {
            } else if constexpr (op.kind == OP_SUBALN) {
                // (one asm statement per dependent pair: between two statements hipcc puts an s_nop for the register they share)
                // (the first four elements of a job start their shift register afresh: four sign bits per register, nothing to mask)
                if constexpr (arg < 4)
                    asm volatile("v_sub_f32 %0, %2, %3\n\tv_lshrrev_b32 %1, 31, %0"
                                 : "=&v"(tt[arg & 3]), "=v"(sh[arg & 3]) : "v"(thr_hi[rb]), "v"(acc[Q][rb][arg]));
                else
                    asm volatile("v_sub_f32 %0, %2, %3\n\tv_alignbit_b32 %1, %1, %0, 31"
                                 : "=&v"(tt[arg & 3]), "+v"(sh[arg & 3]) : "v"(thr_hi[rb]), "v"(acc[Q][rb][arg]));
            } else if constexpr (op.kind == OP_MIN) {
                // pair (arg - 1, arg).  Statements that share a register sit at least two statements apart (hipcc puts an s_nop between
                // closer ones): four rotating t registers and sign-bit shift registers, the minimum two elements behind.
                if constexpr (arg == 1) asm volatile("v_min_u32 %0, %1, %2" : "=v"(mmj[rb]) : "v"(tt[0]), "v"(tt[1]));
                else asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(mmj[rb]) : "v"(tt[(arg - 1) & 3]), "v"(tt[arg & 3]));
            } else if constexpr (op.kind == OP_DSW) {
                // quad `arg` of the job (accumulators 4 arg .. 4 arg + 3) times scale -> the slab, one asm statement: products in a scratch
                // quad (two in turn: the LDS store is still reading its data registers when the next quad's first product is issued
                // right behind it, and nothing interlocks an asm store's sources against an asm VALU write).  scale == 1 (one head, bf16
                // or un-prescaled fp16 operands): the accumulator quad goes to the slab as it is.
                const unsigned wa = slab_w[arg];
                const f32x16& x = acc[Q][rb];
                if constexpr (SCALE1) {
                    const f32x4 q4 = {x[4 * arg], x[4 * arg + 1], x[4 * arg + 2], x[4 * arg + 3]};
                    lds_write128<rb * SLAB_JOB>(wa, q4);
                } else {
                    // (fixed scratch registers named as clobbers: an asm operand cannot address the elements of a register quad)
                    const float sc = scale;
                    if constexpr ((arg & 1) == 0)
                        asm volatile("v_mul_f32 v248, %5, %0\n\tv_mul_f32 v249, %5, %1\n\tv_mul_f32 v250, %5, %2\n\tv_mul_f32 v251, %5, %3\n\t"
                                     "ds_write_b128 %4, v[248:251] offset:%6"
                                     :: "v"(x[4 * arg]), "v"(x[4 * arg + 1]), "v"(x[4 * arg + 2]), "v"(x[4 * arg + 3]), "v"(wa), "s"(sc),
                                        "n"(rb * SLAB_JOB) : "v248", "v249", "v250", "v251", "memory");
                    else
                        asm volatile("v_mul_f32 v252, %5, %0\n\tv_mul_f32 v253, %5, %1\n\tv_mul_f32 v254, %5, %2\n\tv_mul_f32 v255, %5, %3\n\t"
                                     "ds_write_b128 %4, v[252:255] offset:%6"
                                     :: "v"(x[4 * arg]), "v"(x[4 * arg + 1]), "v"(x[4 * arg + 2]), "v"(x[4 * arg + 3]), "v"(wa), "s"(sc),
                                        "n"(rb * SLAB_JOB) : "v252", "v253", "v254", "v255", "memory");
                }
            } else if constexpr (op.kind == OP_CHK) {
                // ONE test per block for both jobs (a ballot + scalar branch costs a lone wave ~60 cycles: 4 % of the launch as one per
                // job).  In the band <=> 0 <= t <= hi - lo; the test here may only be WIDER than the band (laff_rank_resolve applies the
                // exact one to what is dumped).  Rows beyond the matrix have lo = hi = inf: w is NaN, nothing passes.
                const float w0 = (thr_hi[0] - thr_lo[0]) * 1.000002f + 1e-30f, w1 = (thr_hi[1] - thr_lo[1]) * 1.000002f + 1e-30f;
                // (mmj: unsigned minimum of the bits of the job's 16 t values)
                const bool hit0 = w0 >= 0.0f && mmj[0] <= __float_as_uint(w0), hit1 = w1 >= 0.0f && mmj[1] <= __float_as_uint(w1);
#ifdef LAFF_STRIP_NOCHK
                if (false) {
#else
                if (__builtin_amdgcn_ballot_w64(hit0 || hit1) != 0ull) {             // half of the blocks at C4 (fp16 operands)
#endif
                    dump_group(hit0, row_w + l31, cbp * CB + 4 * hh, thr_lo[0], thr_hi[0], 0xffffu, gtc[0], acc[Q][0]);
                    dump_group(hit1, row_w + 32 + l31, cbp * CB + 4 * hh, thr_lo[1], thr_hi[1], 0xffffu, gtc[1], acc[Q][1]);
                }
            } else if constexpr (op.kind == OP_CNT) {
                cnt[rb] += __builtin_popcount(sh[0]) + __builtin_popcount(sh[1]) + __builtin_popcount(sh[2]) + __builtin_popcount(sh[3]);
            } else if constexpr (op.kind == OP_DSR) {
                // (the empty statement keeps rr[arg] allocated from its store to here: hipcc, which does not know that statement is a
                // store, handed the registers to the very next instruction while the store was still reading them)
                asm volatile("" ::"v"(rr[arg]));
                lds_read128<arg * 1024 + rb * SLAB_JOB>(rr[arg], slab_r);
            } else if constexpr (op.kind == OP_WAITR) {
                wait_lgkm<PLAN.wait_r[rb][arg]>();
            } else if constexpr (op.kind == OP_STG) {
#ifndef LAFF_STRIP_NOSTG
                // (locals: a variable named only in an asm operand of a generic lambda is not captured.)  s_nop behind the store: a
                // 16-byte store reads its data registers for a few cycles after issue, hipcc (which does not know this is a store) may
                // hand them to the very next instruction -- seen: an address add landing in lanes 12..15 of the stored data
                const unsigned vo = voffs[4 * rb + arg];
                const u32x4 rs = rsrcSb;
                const u32x4& data = rr[arg];
#ifndef LAFF_STRIP_STFLAVOR
#define LAFF_STRIP_STFLAVOR "sc0 sc1 nt"      // (write-through + streaming: 2 % over "nt", 20 % over plain at C4)
#endif
                asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen " LAFF_STRIP_STFLAVOR :: "v"(data), "v"(vo), "s"(rs) : "memory");
#endif
            }
        };

        // ---- the K loop of one column block (see "the schedule of one column block" above).  PAR = accumulator set of this block; the
        // epilogue of block b - 1 (set PAR ^ 1) rides in the slots.  Ring slots are runtime values (block b lives in slot b % 3).
        // Pieces of blocks beyond the segment read whatever is there (zeros beyond the matrix) into a free slot: no conditions here.
        auto body = [&](auto PARC, int b) {
            constexpr int PAR = decltype(PARC)::value, Q = PAR ^ 1;
            const unsigned sm = (unsigned)b % 3u;
            const unsigned st_cur = lds0 + sm * STAGE;                              // this block's slot: refilled with block b + 3
            const unsigned st_p2 = lds0 + (sm == 0 ? 2u : sm - 1u) * STAGE;         // slot of block b + 2 (= b - 1)
            const unsigned dX = sm == 2 ? (unsigned)(-2 * STAGE) : (unsigned)STAGE; // fragment addresses: this slot -> the next block's
            // descriptors advance by one block per body (a handful of scalar instructions; building them from scratch with 64-bit
            // compares cost 40-90 instructions between the last MFMA of a block and the first of the next)
            const u32x4 rsrcE = rsrcNext;                                           // early pieces: block b + 2
            {
                const unsigned long long nb = (((unsigned long long)rsrcNext.y << 32) | rsrcNext.x) + STAGE;
                rsrcNext.x = __builtin_amdgcn_readfirstlane((unsigned)nb);
                rsrcNext.y = __builtin_amdgcn_readfirstlane((unsigned)(nb >> 32));
                rsrcNext.z = __builtin_amdgcn_readfirstlane(rsrcNext.z > (unsigned)STAGE ? rsrcNext.z - (unsigned)STAGE : 0u);
            }
            const u32x4 rsrcL = rsrcNext;                                           // late pieces: block b + 3
            const int cbp = cb0 + b - 1;
            if constexpr (EPI && HAVE_S) {
                // block b - 1's columns of the strip's score rows; b == 0: there is no previous block, an empty buffer drops the stores
                rsrcSb.x = __builtin_amdgcn_readfirstlane((unsigned)sb_base);
                rsrcSb.y = __builtin_amdgcn_readfirstlane((unsigned)(sb_base >> 32));
                rsrcSb.z = __builtin_amdgcn_readfirstlane(b > 0 ? sb_left : 0u);
                sb_base += CB * 4u;
                sb_left -= b > 0 ? CB * 4u : 0u;
            }
            const unsigned band_addr = lds0 + BAND_OFF + (unsigned)((cb0 + b) >> 1) * 4u;
            float bcv = 0.0f;
            static_for<0, 32>([&](auto JC) {
                constexpr int J = decltype(JC)::value;
                if constexpr (J == BAR_J) {
#if !(LAFF_STRIP_ABL & 2)
                    wait_lgkm<PLAN.lgkm_bar>();
                    wait_vm<PLAN.vm_bar>();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#else
                    wait_lgkm<0>();
#endif
#pragma unroll
                    for (int bb = 0; bb < 8; ++bb) X[bb] += dX;
                    if constexpr (BANDED) {
                        // thresholds of THIS block, used by its epilogue (in the next body or in the drain)
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            const float eps = brow[rb] + bcv;
                            float lo = (sgf[rb] - eps) * inv_scale, hi = (sgf[rb] + eps) * inv_scale;
                            if (row_w + rb * 32 + l31 >= nR) { lo = __builtin_inff(); hi = __builtin_inff(); }   // nothing counted / listed
                            thr_lo[rb] = lo; thr_hi[rb] = hi;
                        }
                    }
                } else if constexpr ((J & 1) == 0) {
#if !(LAFF_STRIP_ABL & 1)
                    wait_lgkm<PLAN.wait_frag[J]>();       // (for sub-steps J and J + 1; the barrier's wait covers BAR_J and BAR_J + 1)
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 2>([&](auto MC) {
                    constexpr int M = decltype(MC)::value, SG = 2 * J + M;
                    mfma16<MODE, J == 0>(acc[PAR][M], fr[J & 7], B[M][J]);
                    __builtin_amdgcn_sched_barrier(0);
                    // pinned fillers
                    constexpr int rs = slot_read_sub(SG), dp = slot_dma_piece(SG);
#if !(LAFF_STRIP_ABL & 1)
                    if constexpr (rs >= 0) lds_read128<((rs & 31) >> 3) * 256>(fr[rs & 7], X[rs & 7]);
#endif
                    if constexpr (BANDED && SG == BAND_READ_SLOT) asm volatile("ds_read_b32 %0, %1" : "=v"(bcv) : "v"(band_addr));
#if !(LAFF_STRIP_ABL & 2)
                    if constexpr (dp >= 2) dma_piece<dp>(lane16x, rsrcE, st_p2 + wrow);
                    else if constexpr (dp >= 0) dma_piece<dp>(lane16x, rsrcL, st_cur + wrow);
#endif
                    // the epilogue stream's share of this slot
                    if constexpr (EPI) {
                        static_for<PLAN.begin[SG], PLAN.begin[SG + 1]>([&](auto IC) {
                            __builtin_amdgcn_sched_barrier(0);
                            epi_item(std::integral_constant<int, Q>{}, IC, cbp);
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        };

        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // The block loop is unrolled by two in straight line (accumulator set = block parity) with ONE instantiation of the body per
        // parity (a separate one for the first block made hipcc copy accumulator sets at the merge): the first body's epilogue runs on
        // the zeroed set 1 with thresholds that count nothing and an empty score buffer.
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[1][0][e] = 0.0f; acc[1][1][e] = 0.0f; }
        {
            int b = 0;
#pragma nounroll
            for (;;) {
                body(I0{}, b);
                if (b < 12) STAMP(tb + 3 + b);
                if (++b >= n) { mfma_drain_nops(); if (EPI) drain_to_lds(I0{}); break; }
                body(I1{}, b);
                if (b < 12) STAMP(tb + 3 + b);
                if (++b >= n) { mfma_drain_nops(); if (EPI) drain_to_lds(I1{}); break; }
            }
            if (EPI) drain_rows(cb0 + n - 1);          // the last block's epilogue is the generic one
        }
        // segment end: nothing of this wave may still be in flight towards LDS or the fragment registers
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) asm volatile("" ::"v"(fr[s8]));
        {
            // the ring restarts at slot 0 with every segment
            const unsigned back = ((unsigned)n % 3u) * STAGE;
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) X[bb] -= back;
        }
        if constexpr (BANDED) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int c = cnt[rb] + __shfl_xor(cnt[rb], 32);
                const int r = row_w + rb * 32 + l31;
                if (hh == 0 && r < nR && c) atomicAdd(a.count + r, c);
            }
        }
        __builtin_amdgcn_s_barrier();            // every wave is done with the ring before the next segment's prologue refills it
        STAMP(tb + 16);
    }
    m0_set(m0_keep);
    if constexpr (BANDED) {
        if (cur_n) flush_chunk();
        if (lane == 0 && cur_chunk < NCH) ((gu32*)pPairs)[4 + cur_chunk] = cur_n;        // close this wave's last chunk
    }
#undef STAMP
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
int g_strip_mode = 1;        // LAFF_STRIP (read when a ctx is created): 0 = never, 1 = where it is the faster kernel (default), 2 = also bf16 operands and
                             // smaller problems, 3 = wherever it can run (score rows of any pitch)
int g_strip_map = 1;         // LAFF_STRIP_MAP: 0 = ranges to workgroups in order, 1 = grouped by column phase per XCD

bool sim_strip_eligible(const GemmArgs& a, int mode, bool aligned) {
    if (g_strip_mode == 0) return false;
    if (mode != GEMM_F16 && mode != GEMM_BF16) return false;
    // bf16 bands are 8 x wider: at C4 nearly every 32 x 32 job has a score inside its band and takes the dump path, which the tiled
    // kernel does faster (0.48 vs 0.68 ms count-only) -- bf16 operands come here only when asked for
    if (mode == GEMM_BF16 && g_strip_mode < 2) return false;
    if (!aligned || a.nseg != 1 || a.K != 512 || a.ldR != 512 || a.ldC != 512) return false;
    if (a.s_gt && !a.s_gt64) return false;                               // the legacy approximate count stays on the tiled kernel
    if (a.count && !a.s_gt64) return false;
    if (a.out && ((a.ldo & 3) || (((uintptr_t)a.out) & 15))) return false;
    if (a.out && (long)SR * a.ldo * 4 >= (1ll << 32)) return false;     // the score rows of a strip are addressed through one raw buffer
    // score rows whose pitch is not a multiple of 64 bytes: the 128-byte row pieces this kernel stores straddle lines at 16 / 32-byte
    // offsets and the launch takes 1.9-2.6 x as long (50000 x 5001: 0.81 ms at ldo 5004 against 0.33 at 5008; tiled 0.46 / 0.39) -- the
    // tiled kernel's 512-byte pieces mind much less.  (laff_amd.ops.alloc_scores pads the pitch to 128 bytes.)
    if (a.out && (a.ldo & 15) && g_strip_mode < 3) return false;
    if ((long)a.nC * KBYTES >= (1ll << 32) || (a.nC + 63) / 64 > STRIP_MAX_GROUPS) return false;
    if (a.pairs && (16ull + 8ull * a.pair_cap) >= (1ull << 32)) return false;
    const long units = (long)((a.nR + SR - 1) / SR) * ((a.nC + CB - 1) / CB);
    // measured against the tiled kernel (fp16, count-only / with scores): 40 x 63 units 0.043 / 0.058 vs 0.036 / 0.050 ms, 20 x 157 equal /
    // 0.069 vs 0.063, 79 x 125 0.097 / 0.135 vs 0.107 / 0.140, 157 x 313 (C4) 0.359 / 0.496 vs 0.464 / 0.571, 391 x 313 0.858 / 1.176 vs 1.124 / 1.387
    // (round 4, with the staged strip load, `tools/debug/time_strip_small.py`: 40 x 94 units [14.7 per CU] 52 / 69 us vs 56 / 63 tiled;
    // 79 x 125 [38.6] 92 / 122 vs 98 / 131; 234 x 94 [86] 167 / 223 vs 223 / 267.  Without the score stores the strip form would win
    // from ~12 units per CU on, but its dump list takes 96 bytes per hit lane: at C3 -- chance-level scores, 1 % of the pairs inside
    // the band -- the default list overflows where the tiled kernel's 8-byte pairs fit.  One threshold for both modes.)
    return units >= (g_strip_mode >= 2 ? 8L : 24L) * g_num_cus;
}

hipError_t launch_sim_strip(const GemmArgs& a, int mode, hipStream_t st) {
    StripArgs s{};
    s.T = a.R; s.V = a.C; s.nR = a.nR; s.nC = a.nC; s.out = a.out; s.ldo = a.ldo; s.scale = a.scale; s.inv_scale = 1.0f / a.scale;
    s.gt_col = a.gt_col; s.col0 = a.col0; s.s_gt64 = a.s_gt64; s.band_r = a.band_r; s.band_c = a.band_c; s.count = a.count;
    s.pairs = a.pairs; s.pair_cap = a.pair_cap;
    s.debug = g_strip_mode;
#ifdef LAFF_STRIP_TRACE
    if (const char* e = getenv("LAFF_GEMM_TRACE_PTR")) s.trace = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
    const int NB = (a.nC + CB - 1) / CB, NS = (a.nR + SR - 1) / SR;
    const long U = (long)NS * NB;
    int G = (int)std::min<long>(std::min(g_num_cus, STRIP_MAX_WG), U);
    s.nranges = G;
    // ranges -> workgroups.  Workgroup b runs on XCD b % 8 (observed; only speed depends on it): the ranges are sorted by the column
    // phase they start at and dealt out so that the 1/8 of them closest in phase share an XCD -- the CUs of an XCD then walk the
    // video operand within a window of NB / 8 column blocks and their L2 fetches every video row once per sweep.
    std::vector<int> order(G);
    for (int j = 0; j < G; ++j) order[j] = j;
    if (g_strip_map == 1 && G % 8 == 0) {
        std::vector<long> phase(G);
        for (int j = 0; j < G; ++j) phase[j] = (U * j / G) % NB;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return phase[x] < phase[y]; });
        for (int b = 0; b < G; ++b) s.range_of_wg[b] = (unsigned short)order[(b % 8) * (G / 8) + b / 8];
    } else {
        for (int b = 0; b < G; ++b) s.range_of_wg[b] = (unsigned short)b;
    }
    const bool banded = a.s_gt64 != nullptr && a.count != nullptr;
    const bool hs = a.out != nullptr;
#define LAFF_STRIP_LAUNCH(M, BD, HS, S1)                                                                                      \
    do {                                                                                                                      \
        static unsigned long long attr_done = 0;                                                                              \
        if (hipError_t e = smem_attr_once(attr_done, sim_strip_kernel<M, BD, HS, S1>, SMEM); e != hipSuccess) return e;       \
        hipLaunchKernelGGL((sim_strip_kernel<M, BD, HS, S1>), dim3((unsigned)G), dim3(256), SMEM, st, s);                     \
    } while (0)
#define LAFF_STRIP_MODE(M)                                                                                                    \
    do {                                                                                                                      \
        if (banded && hs && s1) LAFF_STRIP_LAUNCH(M, true, true, true);                                                       \
        else if (banded && hs) LAFF_STRIP_LAUNCH(M, true, true, false);                                                       \
        else if (banded) LAFF_STRIP_LAUNCH(M, true, false, false);                                                            \
        else if (s1) LAFF_STRIP_LAUNCH(M, false, true, true);                                                                 \
        else LAFF_STRIP_LAUNCH(M, false, true, false);                                                                        \
    } while (0)
    const bool s1 = a.scale == 1.0f;
    if (mode == GEMM_F16) LAFF_STRIP_MODE(GEMM_F16);
    else LAFF_STRIP_MODE(GEMM_BF16);
#undef LAFF_STRIP_MODE
#undef LAFF_STRIP_LAUNCH
    return hipGetLastError();
}

}  // namespace laff
