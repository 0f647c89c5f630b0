// kernels.h -- internal launch interface between api.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/laff_hip.h"

namespace laff {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: one flag word per kernel instantiation, one bit per device
// ordinal (a second laff_ctx on another GPU of the same process sets it again).  The first launch of an instantiation on a device must
// therefore happen outside a HIP-graph capture (every caller warms up eagerly before it captures).
template <typename K>
static inline hipError_t smem_attr_once(unsigned long long& done, K kernel, int smem) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    // (atomic: two host threads driving two contexts must not lose each other's bit -- a lost bit would repeat the call, possibly
    // inside a capture)
    if (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    if (smem > 64 * 1024) {
        e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
    }
    __atomic_fetch_or(&done, bit, __ATOMIC_ACQ_REL);
    return hipSuccess;
}

enum { GEMM_F32 = 0, GEMM_F16 = 1, GEMM_BF16 = 2 };

struct GemmArgs {
    const void* R;        // [nR, ldR] rows -> output rows
    const float* Rf;      // x3 fused tile only: the row operand in fp32 [nR, ldRf]; split into hi/lo planes in the kernel
    int ldRf;
    const void* C;        // [nC, ldC] rows -> output columns
    int nR, nC, K;        // K in elements (per segment)
    int ldR, ldC;         // elements
    int nseg;             // 1, or 3 for the hi/lo split (virtual K concatenation)
    long segR[3], segC[3];  // byte offset of the operand plane used by each segment
    float* out;           // [nR, ldo] or null
    int ldo;
    float scale;
    const float* row_scale;   // optional per-output-row / per-output-column factors (operand split scales)
    const float* col_scale;
    // FC epilogue
    const float* bias;
    const float* bn_scale;
    const float* bn_shift;
    int act;
    // fused ground-truth rank count
    const int* gt_col;
    int col0;
    const float* s_gt;
    int* count;
    // exact-rank ("banded") count: definite counts here, pairs inside the error band go to a list for laff_rank_resolve
    const double* s_gt64;     // [nR] exact ground-truth scores (laff_rank_prepare); non-null selects the banded epilogue
    const float* band_r;      // [nR] row part of the band half-width
    const float* band_c;      // [nC] column part
    unsigned* pairs;          // header {count, overflow, 0, 0} + pair_cap x {row, col}
    unsigned pair_cap;
    unsigned long long* trace;   // debug (LAFF_GEMM_TRACE build only): 8 timestamps per workgroup
};

constexpr int MAX_GROUP = 8;
struct GroupedGemmArgs {
    int count;
    int nbig;             // x3 grouped kernel: blocks [0, nbig) run big tiles, the rest quarter tiles of the remaining ones
    int tile_start[MAX_GROUP + 1];
    GemmArgs p[MAX_GROUP];
};

// ---- strip form of the similarity GEMM at K = 512 (sim_strip.hip) ------------------------------------------------------------
constexpr int STRIP_ROWS = 256;          // text rows per strip (4 wavefronts x 64 rows held in registers)
constexpr int STRIP_COLS = 32;           // videos per column block (one 32 KiB ring slot)
constexpr int STRIP_MAX_GROUPS = 2048;   // aligned 64-column groups whose band maxima fit the LDS table (131,072 videos)
constexpr int STRIP_MAX_WG = 512;        // persistent workgroups (one per CU)
constexpr int STRIP_CHUNK = 64;          // entries per chunk of the dump list (a wavefront takes a new chunk with one atomic)
constexpr int STRIP_ENTRY_WORDS = 24;    // one dumped group: {row, colbase, lo, hi | mask16, the row's ground-truth column, 0, 0 | 16 raw accumulators}
struct StripArgs {
    const void* T;            // [nR][512] 16-bit, rows 1,024 bytes apart
    const void* V;            // [nC][512]
    int nR, nC;
    float* out;               // [nR][ldo] or null
    int ldo;
    float scale, inv_scale;
    const int* gt_col;        // banded count (as GemmArgs); count == null: scores only
    int col0;
    const double* s_gt64;
    const float* band_r;
    const float* band_c;
    int* count;
    unsigned* pairs;          // header {chunks taken, flag, NW | 1 << 31, NCH} | NCH per-chunk counts | NCH chunks of STRIP_CHUNK entries
    unsigned pair_cap;
    int nranges;              // == gridDim.x
    int debug;                // g_strip_mode (3: K loops only)
    unsigned long long* trace;   // debug (LAFF_STRIP_TRACE build only): 64 cycle stamps per workgroup
    unsigned short range_of_wg[STRIP_MAX_WG];
};
extern int g_strip_mode, g_strip_map;
bool sim_strip_eligible(const GemmArgs& a, int mode, bool aligned);
hipError_t launch_sim_strip(const GemmArgs& a, int mode, hipStream_t st);

// ---- strip form of the FC projection at K = 512 (fc_strip.hip): X stationary in registers, W streamed from a packed LDS image ----
constexpr int FC_STRIP_ROWS = 128;       // input rows per strip (4 wavefronts x 32 rows held in registers)
constexpr int FC_STRIP_K = 512;
struct FcStripProblem {
    const float* X;           // [N][ldx] fp32, 16-byte aligned rows
    const void* img;          // laff_fc_strip_pack: D / 32 blocks x 2 K-halves x 32 KiB {hi, lo} + [D][4] lane constants
    const float* vec;         // the lane constants inside img
    float* Y;                 // [N][ldy]
    int ldx, ldy, N;
    int unit0;                // first (strip, column block) unit of this problem in the launch (filled by launch_fc_strip)
};
struct FcStripArgs {
    int count;
    int nblk;                 // D / 32, the same for every problem of a launch
    int nranges;              // == gridDim.x
    int total_units;
    unsigned long long* trace;   // debug (LAFF_FCS_TRACE build only): 80 words per workgroup
    FcStripProblem p[MAX_GROUP];
    unsigned short range_of_wg[STRIP_MAX_WG];
};
extern int g_fc_strip;
size_t fc_strip_image_bytes(int D);
size_t fc_strip_vec_offset(int D);
hipError_t launch_fc_strip_pack(const float* W, int ldw, const float* bias, const float* bn_scale, const float* bn_shift, int D, int act,
                                void* img, hipStream_t st);
hipError_t launch_fc_strip(FcStripArgs& a, int act, hipStream_t st);

hipError_t launch_gemm_nt(const GemmArgs& a, int mode, bool aligned, hipStream_t st);
hipError_t launch_gemm_nt_grouped_f32(GroupedGemmArgs& g, int staging, hipStream_t st);
hipError_t launch_gemm_nt_grouped_f16(GroupedGemmArgs& g, hipStream_t st);
hipError_t launch_gemm_nt_x3_fused_grouped(GroupedGemmArgs& g, hipStream_t st);   // fp32 row operand split in the kernel   // fast staging only (packed operands)
int staging_kind(const GemmArgs& a, int esz, bool aligned);
extern int g_gemm_variant;
extern int g_num_cus;

constexpr int MAX_L = 8;
struct FuseArgs {
    const float* src[MAX_L];
    const float* scale[MAX_L];
    const float* shift[MAX_L];
    int ld[MAX_L];
    int tile[MAX_L];
    int act[MAX_L];       // LAFF_ACT_* applied to the plane before its affine
    const float* rownorm[MAX_L];      // optional per-row factor [N] applied after the affine (expert-embedding l2norm branch)
    // gather planes (src == null): plane value = sum_j values_j * Wt[indices_j, column] + bias[column]  (sparse bag-of-words FC)
    const int* g_indptr[MAX_L];
    const int* g_indices[MAX_L];
    const float* g_values[MAX_L];     // null = all ones
    const float* g_wt[MAX_L];         // [Dk, g_ldwt], non-null marks a gather plane
    const float* g_bias[MAX_L];
    int g_ldwt[MAX_L], g_dk[MAX_L];
    int head_major;       // block -> (head = b % H, 4 rows): with H == 8 every XCD gathers from its own 512-column slice of Wt
    int L, N, H, d;
    int head_stride;      // d (split heads) or 0 (every head sees all columns)
    const float* w;       // [H, d]
    const float* b;       // [H]
    const float* gw;      // [H]
    unsigned flags;
    float* E;             // [N, H, d]
    float* attn_w;        // [N, H, L] or null
    void* E16;            // optional [N, H*d] 16-bit GEMM operand (E * e16_scale), fp16 or bf16
    int e16_bf16;
    float e16_scale;
    // laff_rank_prepare's work for these rows done by this launch (laff_fuse_packed_rank; heads of d <= 512, E16 given): the wave
    // that has just produced a row measures its operand's rounding error (band) and, on the text side, scores it exactly against its
    // ground-truth video (s_gt64) -- the rows are not read back by a separate launch.  rp_side 0 = off, 1 = text rows, 2 = video rows.
    int rp_side;
    const int* rp_gt;         // side 1: [N] ground-truth column of every text
    int rp_col0, rp_Nv;       // side 1: the videos of this launch's partner are columns [col0, col0 + Nv)
    const float* rp_Ev;       // side 1: their fp32 embeddings [Nv, d] (the E of the side-2 launch, complete before this one starts)
    double* rp_sgt;           // side 1: s_gt64 [N]
    float* rp_band;           // side 1: band_t [N];  side 2: band_v [((N + 3) & ~3) + ceil(N / 64)] (per column; the block maxima are
                              //         finished by the side-1 launch, which runs behind this one)
    float* rp_band_v;         // side 1: the partner's band_v
    int* rp_count;            // side 1: [N], cleared
    unsigned* rp_pairs;       // side 1: pair-list header, cleared
    double* rp_part;          // H > 1: [N][H][2] scratch {head term of the exact score, q_h^2}
    unsigned* rp_ticket;      // H > 1: [N] arrival counters, zero at launch
    float rp_unit, rp_cacc;   // unit roundoff of the operand format, accumulation term of the band (see launch_rank_prepare)
};
hipError_t launch_fuse(const FuseArgs& a, hipStream_t st);
hipError_t launch_plane_row_norms(const FuseArgs& a, float* out /*[L][N]*/, hipStream_t st);

struct FrameArgs {
    const float* frames;  // [B, Fmax, d]
    const int* lens;      // [B] or null
    int B, Fmax, d;
    const float* w;
    const float* b;
    const float* gw;
    unsigned flags;
    float* V;             // [B, d]
    const float* mask;    // [B, ldm] or null: the reference's mask_tensor (1.0 per valid frame, a prefix of the row); replaces lens
    int ldm;
};
struct FrameGroup {       // up to 8 frame features of the same shape in ONE launch: block -> (feature, video)
    int count;
    FrameArgs f[8];
};
hipError_t launch_frame_fuse(const FrameGroup& g, hipStream_t st);

hipError_t launch_split_rows_grouped(int count, const float* const* X, const int* N, const int* K, const int* ldx, void* const* out,
                                     float* const* rscale, hipStream_t st);
hipError_t launch_split_rows(const float* X, int N, int K, int ldx, int Kp, void* out, float* rscale, hipStream_t st);
hipError_t launch_loss_normalize(const float* s, const float* im, int B, int H, int d, int dp, int Bp, float eps, float* XH,
                                 float* XHT, float* nrm, float* npr, hipStream_t st);
hipError_t launch_margin_reduce(const float* S, float* dS, float* dST, float* loss_h, float* loss, int B, int Bp, int H,
                                float margin, int max_violation, int use_s, int use_im, float g_s, float g_im, hipStream_t st);
hipError_t launch_loss_normalize_bwd(const float* XH, const float* G, const float* nrm, const float* npr, int B, int H, int d,
                                     int dp, float* d_s, float* d_im, hipStream_t st);
hipError_t launch_fc_gather(const int* indptr, const int* indices, const float* values, int N, int Dk, const float* Wt, int ldwt,
                            const float* bias, const float* bn_scale, const float* bn_shift, int D, int act, float* Y, int ldy,
                            hipStream_t st);
hipError_t launch_pack_rows(const float* E, int N, int H, int d, int lde, int normalize, float eps, float prescale,
                            int precision, void* out, hipStream_t st);

hipError_t launch_row_dot_gt(const void* T, const void* V, int Nt, int Nv, int K, int bf16, int x3, float scale,
                             const int* gt_col, int col0, float* s_gt, int* zero_count, hipStream_t st);
hipError_t launch_rank_prepare(const float* Et, const float* Ev, const void* T, const void* V, int Nt, int Nv, int H, int d,
                               int precision, float prescale, const int* gt_col, int col0, double* s_gt64, float* band_t,
                               float* band_v, int* zero_count, unsigned* pairs, int sides, hipStream_t st, int emit = 0);
hipError_t launch_rank_export(const double* s_gt64, int* count, float* S, int lds, unsigned* pairs, unsigned pair_cap, const int* bounds,
                              int world, int col0, unsigned* out, unsigned cap, unsigned* fill, hipStream_t st);
// metrics_n > 0: the block that finishes last also turns the counts into ranks (count + base -> ranks_out) and the seven metrics
// (out8 on the device, host8 = the same in device-addressable host memory or null); ticket: one zero word, left zero
hipError_t launch_rank_resolve(const float* Et, const float* Ev, int Nt, int Nv, int H, int d, const double* s_gt64, int* count,
                               float* S, int lds, unsigned* pairs, unsigned pair_cap, hipStream_t st, int metrics_n = 0, int base = 0,
                               int* ranks_out = nullptr, double* out8 = nullptr, double* host8 = nullptr, unsigned* ticket = nullptr);
// scratch: rank_metrics_scratch_bytes() bytes, zero before the first launch (every launch leaves it zero); not shared by launches in flight
size_t rank_metrics_scratch_bytes();
size_t rank_resolve_ticket_offset();      // where, inside that scratch, the ticket lines of the fused resolve + metrics launch start
// host8: the 8 result doubles also go to this device-addressable host buffer (null: none)
hipError_t launch_rank_metrics(const int* r, int n, int base, int* ranks_out, double* out7, double* err, unsigned* scratch, hipStream_t st,
                               double* host8 = nullptr);
hipError_t launch_gather_gt(const float* S, int Nt, int Nv, int lds, const int* gt_col, int col0, float* s_gt,
                            hipStream_t st);
hipError_t launch_rank_count(const float* S, int Nt, int Nv, int lds, const int* gt_col, int col0, const float* s_gt,
                             int* count, int accumulate, hipStream_t st);
hipError_t launch_topk_rows(const float* S, int Nt, int Nv, int lds, int K, int* idx_out, float* val_out, hipStream_t st);
hipError_t launch_v2t_count(const float* S, int Nt, int Nv, int lds, const int* grp_off, const int* grp_idx,
                            int max_group, int* count, hipStream_t st);
hipError_t launch_v2t_count_exact(const float* S, int Nt, int Nv, int lds, const int* grp_off, const int* grp_idx, int max_group,
                                  const float* Et, const float* Ev, int H, int d, const double* s_gt64, const float* band_t,
                                  const float* band_v, int* count, unsigned* list, unsigned cap, hipStream_t st);

}  // namespace laff
