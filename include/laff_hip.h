/*
 * laff_hip.h -- C ABI of liblaff_hip.so: the MI355X (gfx950) kernels behind the LAFF retrieval hot path.
 *
 * The reference (ruc-aimc-lab/LAFF, pure Python/PyTorch) has no FFI layer; its boundary is the nn.Module
 * API in model/model.py.  laff_amd/ keeps that Python surface and calls the entry points below through
 * ctypes.  Each entry point names the reference code it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - every pointer is CALLER-OWNED DEVICE MEMORY (hipMalloc / a torch tensor's data_ptr) unless the
 *     parameter is documented as host memory; the library neither frees nor retains it past the call;
 *   - all matrices are row-major fp32 unless stated; `ld*` = leading dimension in ELEMENTS;
 *   - calls are asynchronous and ordered on the stream bound to the ctx (laff_ctx_set_stream);
 *     only laff_rank_metrics / laff_device_info synchronise;
 *   - return value: 0 = LAFF_OK, negative = error; laff_last_error() returns the thread-local message;
 *   - an EMPTY problem (N = 0 rows / B = 0 videos / Nt = 0 or Nv = 0) is legal everywhere -- the reference meets it as an
 *     empty last batch or an empty query set: the call returns LAFF_OK without launching or touching any pointer;
 *   - a ctx is not thread-safe; distinct ctxs are independent.  No exceptions cross this boundary.
 */
#ifndef LAFF_HIP_H
#define LAFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAFF_ABI_VERSION 22

enum {
    LAFF_OK = 0,
    LAFF_E_ARG = -1,          /* null pointer / negative size / bad enum */
    LAFF_E_SHAPE = -2,        /* shape outside what the kernels support (message says which) */
    LAFF_E_ALIGN = -3,        /* pointer / leading dimension alignment */
    LAFF_E_HIP = -4,          /* a HIP runtime call failed (message carries hipGetErrorString) */
    LAFF_E_UNSUPPORTED = -5
};

/* activation of TransformNet (model/model.py:236-243) */
enum { LAFF_ACT_NONE = 0, LAFF_ACT_TANH = 1, LAFF_ACT_RELU = 2, LAFF_ACT_SIGMOID = 3 };

/* flags of the attention kernels (Attention_1 ctor, model/Attention.py:48-63; Multi_head ctor :489-506) */
enum {
    LAFF_ATT_WITH_AVE = 1,          /* g += gw * mean_L(x)           (Attention.py:94-99) */
    LAFF_ATT_MUL = 2,               /* logits on x * mean_L(x)       (Attention.py:83-86) */
    LAFF_ATT_L2NORM_EACH_HEAD = 4,  /* l2norm(x, dim=3) per head     (Attention.py:522-523) */
    LAFF_ATT_NO_SPLIT_HEAD = 8,     /* every head sees all D columns (Attention.py:520)   */
    LAFF_ATT_JUST_AVERAGE = 16      /* JustAverage: mean over L, no softmax, no norm (Attention.py:35-37) */
};

/* operand precision of the similarity GEMM */
enum {
    LAFF_PREC_FP32 = 0,    /* fp32 MFMA (v_mfma_f32_32x32x2_f32): bit-for-bit an fp32 fma chain          */
    LAFF_PREC_FP16 = 1,    /* one fp16 MFMA pass: MEASURED max |d cos| 8e-5 on 4e8 pairs at d=512; the PROVEN per-pair bound is
                            * band_t + band_v of laff_rank_prepare, 4-5e-4 at d=512 (Cauchy-Schwarz on the operand rounding).  Ranks
                            * from the banded pipeline are exact regardless; a caller that needs every returned score inside 1e-4
                            * takes FP16X3 (what predict() / retrieve() default to) */
    LAFF_PREC_BF16 = 2,    /* one bf16 MFMA pass: ~5e-4 -- for rank-identity workloads only              */
    LAFF_PREC_FP16X3 = 3,  /* fp16 hi+lo split, 3 MFMA passes: ~1e-7                                    */
    LAFF_PREC_BF16X3 = 4   /* bf16 hi+lo split, 3 MFMA passes: ~1e-6                                    */
};

typedef struct laff_ctx laff_ctx;

/* ---- context ------------------------------------------------------------------------------------ */
int laff_abi_version(void);
const char* laff_last_error(void);
/* device: HIP ordinal; hip_stream: hipStream_t (NULL = default stream). */
int laff_ctx_create(int device, void* hip_stream, laff_ctx** out);
int laff_ctx_set_stream(laff_ctx* ctx, void* hip_stream);
int laff_ctx_destroy(laff_ctx* ctx);
/* host out: [0]=CU count, [1]=clock MHz, [2]=LDS bytes per CU-workgroup limit, [3]=wavefront size */
int laff_device_info(laff_ctx* ctx, int out[4]);
/* Measurement aid (no counterpart in the reference): a one-thread launch, ordered on the ctx's stream like every other call and
 * capturable in a HIP graph, that writes the device's constant-rate wall clock (s_memrealtime) to *slot (device memory).  bench.py puts
 * one in front of and behind every launch of a captured step: the differences are the durations of the kernels as the graph runs
 * them (this runtime refuses event-record nodes in a capture).  laff_wall_clock_khz: ticks per millisecond of that clock. */
int laff_stamp(laff_ctx* ctx, unsigned long long* slot /*device*/);
int laff_wall_clock_khz(laff_ctx* ctx, int* khz /*host*/);

/* ---- a1: TransformNet.forward (model/model.py:257-276), eval mode ------------------------------------
 * Y[N,D] = (act(X[N,Dk] . W[D,Dk]^T + bias)) * bn_scale + bn_shift
 * bias / bn_scale / bn_shift may be NULL (absent stage).  bn_* are the folded eval-mode BatchNorm1d:
 * scale = gamma / sqrt(running_var + 1e-5), shift = beta - running_mean * scale.
 * fp32 MFMA; any N, Dk, D >= 1 (16-byte aligned rows take the direct-to-LDS path). */
int laff_fc_act_bn(laff_ctx* ctx, const float* X, int N, int Dk, int ldx, const float* W, int ldw,
                   const float* bias, const float* bn_scale, const float* bn_shift, int D, int act,
                   float* Y, int ldy);

/* The same projection for up to 8 independent features in ONE launch (a3: the per-feature Python loop of
 * VisMutiTransformNet.forward, model/model.py:1807-1827, and the text-side loop :1673-1681). */
typedef struct {
    const float* X; int N, Dk, ldx;
    const float* W; int ldw;
    const float* bias; const float* bn_scale; const float* bn_shift;
    int D, act;
    float* Y; int ldy;
} laff_fc_problem;
int laff_fc_act_bn_grouped(laff_ctx* ctx, const laff_fc_problem* problems /*host array*/, int count);

/* a1 on the 16-bit matrix pipe with fp32-class accuracy ("fp16x3"): both operands are split once into exact fp16
 * hi + lo parts with a per-row power-of-two scale (laff_split_rows); the GEMM accumulates lo*hi + hi*lo + hi*hi in fp32
 * (fp16 x fp16 products are exact in fp32; the dropped lo*lo term is 2^-22 relative) and the epilogue undoes the scales.
 * 3 MFMA passes at the 2.5 PF fp16 rate instead of one at the 157 TF fp32 rate. */
int laff_split_rows_bytes(int N, int K, size_t* out);            /* bytes of the [2][N][Kp] fp16 operand, Kp = ceil(K/64)*64 */
int laff_split_rows(laff_ctx* ctx, const float* X, int N, int K, int ldx, void* out_hi_lo, float* rscale /*[N]*/);
/* up to 8 matrices in one launch (host arrays of `count` entries) */
int laff_split_rows_grouped(laff_ctx* ctx, int count, const float* const* X, const int* N, const int* K, const int* ldx,
                            void* const* out_hi_lo, float* const* rscale);
typedef struct {
    const void* Xs; const float* x_rscale; int N, Dk;          /* laff_split_rows(X[N,Dk]) */
    const void* Ws; const float* w_rscale;                      /* laff_split_rows(W[D,Dk]) */
    const float* bias; const float* bn_scale; const float* bn_shift;
    int D, act;
    float* Y; int ldy;
} laff_fc_split_problem;
int laff_fc_act_bn_split_grouped(laff_ctx* ctx, const laff_fc_split_problem* problems /*host array*/, int count);

/* The same projection with the INPUT split fused into the GEMM: X stays fp32 in HBM, only its per-row power-of-two scales are
 * computed beforehand (laff_row_scales_grouped: one read of X, N floats out; same scales as laff_split_rows) and the kernel
 * forms the fp16 hi / lo planes on the way into LDS -- laff_split_rows' 2 x N x Kp fp16 planes are never written or re-read.
 * Results are bit-identical to laff_split_rows + laff_fc_act_bn_split_grouped on 256x256 tiles.
 * Needs Dk % 32 == 0, ldx % 4 == 0 and 16-byte aligned X rows; W is split once with laff_split_rows as before. */
int laff_row_scales_grouped(laff_ctx* ctx, int count, const float* const* X, const int* N, const int* K, const int* ldx,
                            float* const* rscale);
typedef struct {
    const float* X; int ldx; const float* x_rscale; int N, Dk;  /* fp32 input + laff_row_scales_grouped(X) */
    const void* Ws; const float* w_rscale;                      /* laff_split_rows(W[D,Dk]) */
    const float* bias; const float* bn_scale; const float* bn_shift;
    int D, act;
    float* Y; int ldy;
} laff_fc_fused_problem;
int laff_fc_act_bn_fused_grouped(laff_ctx* ctx, const laff_fc_fused_problem* problems /*host array*/, int count);

/* a1 / a3 in STRIP form for Dk == 512 inputs (TransformNet.forward, model/model.py:257-276; the per-feature loops :1807-1827 and
 * :1673-1681): same arithmetic as the fused-split form above (fp16 hi/lo, three products, fp32 accumulation) with X STATIONARY --
 * a wavefront loads 32 fp32 input rows once, finds their maxima and converts them to hi / lo MFMA fragments in registers (no
 * laff_row_scales_grouped pass, nothing of the split ever in memory); W streams through LDS from an image packed once per model:
 *   laff_fc_strip_pack(W[D,512], bias, bn_scale, bn_shift, act) -> img  (laff_fc_strip_pack_bytes(D) bytes, 16-byte aligned: the hi/lo
 *   fragments of W in LDS order + the per-column epilogue constants with bias, activation and folded BatchNorm pre-combined)
 * Needs Dk == 512, D % 32 == 0, ldx % 4 == 0, 16-byte aligned X; any N, any ldy >= D.  Up to 8 problems per launch; problems of
 * different D or activation kind are launched separately.  |Y - fp64| as the other fp16x3 forms (<= ~1e-6 of the row / weight scale). */
int laff_fc_strip_pack_bytes(int D, int Dk, size_t* out);
int laff_fc_strip_pack(laff_ctx* ctx, const float* W, int ldw, const float* bias, const float* bn_scale, const float* bn_shift, int D,
                       int Dk, int act, void* img);
typedef struct {
    const float* X; int ldx; int N;
    const void* img; int D, act;                                /* laff_fc_strip_pack(...) of this feature's TransformNet */
    float* Y; int ldy;
} laff_fc_strip_problem;
int laff_fc_act_bn_strip_grouped(laff_ctx* ctx, const laff_fc_strip_problem* problems /*host array*/, int count);

/* a1 for a SPARSE input feature (bag-of-words, model/model.py:399-416): X given as CSR (indptr[N+1], indices[nnz],
 * values[nnz] or NULL = all ones) over Dk columns; Wt = W^T [Dk, ldwt >= D] so that one vocabulary entry is one contiguous
 * row.  Y[N,D] = bn(act(sum_j values_j * Wt[indices_j, :] + bias)).  D % 4 == 0, D <= 8192, 16-byte aligned Wt / Y rows. */
int laff_fc_gather_act_bn(laff_ctx* ctx, const int* indptr, const int* indices, const float* values, int N, int Dk,
                          const float* Wt, int ldwt, const float* bias, const float* bn_scale, const float* bn_shift, int D,
                          int act, float* Y, int ldy);

/* ---- 8f-4: training loss -- MarginRankingLoss (loss.py:68-135) per head, summed over heads (model/model.py:2032-2048) ----
 * s = caption embeddings, im = video embeddings, both [B, H, d] fp32 contiguous (H = 1 for a 2-D batch).
 * scores_h = l2norm(im_h) . l2norm(s_h)^T (eps 1e-13, loss.py:30-34); hinge on the margin against the diagonal, row-wise
 * ('i2t'), column-wise ('t2i') or both; max_violation keeps the hardest negative; cost_style sum or mean.
 * Writes loss[0] (device float) and, when non-NULL, the gradients d_s / d_im [B, H, d] of that loss (what autograd gives the
 * reference).  workspace: caller-owned device scratch of laff_margin_loss_workspace_bytes(B, H, d), 16-byte aligned. */
enum { LAFF_LOSS_MAX_VIOLATION = 1, LAFF_LOSS_COST_MEAN = 2, LAFF_LOSS_DIR_I2T = 4, LAFF_LOSS_DIR_T2I = 8 };
int laff_margin_loss_workspace_bytes(int B, int H, int d, size_t* out);
int laff_margin_loss(laff_ctx* ctx, const float* s, const float* im, int B, int H, int d, float margin, unsigned flags,
                     float* loss, float* d_s, float* d_im, void* workspace, size_t workspace_bytes);

/* ---- a2-a6: stack + Multi_head_MyApply_Attention / Attention_1 / JustAverage ----------------------------
 * (model/model.py:1858-1876, :1663-1705; model/Attention.py:508-531, :78-105)
 * One feature plane per fused feature; nothing is stacked or tiled in memory.
 *   x_l[n, h, c] = act_l( src_l[n, (tile ? c : h*d + c)] ) * scale_l[h*d + c] + shift_l[h*d + c]
 * act (LAFF_ACT_*) lets the producing projection hand over its PRE-activation output (x W^T + b) and leave
 * `tanh -> BatchNorm` of TransformNet (model/model.py:270-275) to this kernel (same v_exp_f32 / v_rcp_f32 arithmetic as the
 * GEMM epilogue: identical bits either way).
 * tile != 0 restates the no-transform branch `x.repeat(1, heads)` + BatchNorm1d(D)
 * (model/model.py:1801-1805, 1822-1823; text side :659-664, 1675-1676): src has d columns. */
typedef struct {
    const float* src;     /* [N, ld] */
    int ld;
    int tile;
    const float* scale;   /* [H*d] or NULL (=1) */
    const float* shift;   /* [H*d] or NULL (=0) */
    int act;              /* LAFF_ACT_* applied to src before the affine (0 = none) */
    /* GATHER plane (src == NULL, wt != NULL): a sparse feature through its FC without materialising the projected plane --
     * value[n, c] = sum_j values[j] * wt[indices[j], c] + bias[c] over the CSR row n (the arithmetic of laff_fc_gather_act_bn),
     * then act and the affine as above.  d <= 512 per head, never tiled. */
    const int* indptr;    /* [N+1] */
    const int* indices;   /* [nnz] */
    const float* values;  /* [nnz] or NULL (all ones) */
    const float* wt;      /* [dk, ldwt] = W^T */
    int ldwt, dk;
    const float* bias;    /* [H*d] or NULL */
    /* optional per-row factor [N] applied after the affine: the inverse row norm of `local_embs = l2norm(local_embs, dim=2)` in the
     * expert-embedding branch (model/model.py:1866-1873, :1686-1694), produced by laff_plane_row_norms; NULL = none.  Not with gather
     * planes. */
    const float* row_scale;
} laff_plane;

/* out[l * N + n] = 1 / (|x_l[n, :]|_2 + 1e-13 + 1e-14) over all H*d columns of plane l as laff_fuse would read it (activation,
 * tiling, affine -- the expert embedding row rides in `shift`): what `l2norm(local_embs, dim=2)` divides by (loss.py:8-13).
 * The planes' own row_scale must be NULL here.  flags: only LAFF_ATT_NO_SPLIT_HEAD matters (d = D columns, one "head"). */
int laff_plane_row_norms(laff_ctx* ctx, const laff_plane* planes /*host array of L*/, int L, int N, int H, int d, unsigned flags,
                         float* out /*[L, N]*/);

/* E[N,H,d] (unit L2 norm per (n,h) unless JUST_AVERAGE).  w [H,d], b [H], gw [H] device arrays.
 * attn_w: optional [N,H,L] softmax weights (the `self.weights` side output, Attention.py:90).
 * L <= 8, d % 4 == 0.  With NO_SPLIT_HEAD every head reads columns [0,d) (d = D). */
int laff_fuse(laff_ctx* ctx, const laff_plane* planes /*host array of L*/, int L, int N, int H, int d,
              const float* w, const float* b, const float* gw, unsigned flags, float* E, float* attn_w);

/* laff_fuse that ALSO emits the single-plane 16-bit similarity operand E16[N, H*d] = E * prescale (LAFF_PREC_FP16 or
 * LAFF_PREC_BF16), saving the separate laff_pack_rows pass.  E is already unit-norm per (n,h) to fp32 rounding, so the
 * re-normalisation loss.cosine_sim applies (loss.py:33) is a no-op at 16-bit precision. */
int laff_fuse_packed(laff_ctx* ctx, const laff_plane* planes, int L, int N, int H, int d, const float* w, const float* b,
                     const float* gw, unsigned flags, float* E, float* attn_w, void* E16, int precision, float prescale);

/* laff_fuse_packed that ALSO does laff_rank_prepare's work for its rows (split heads of d <= 512): the wavefront that has just produced a
 * row measures the rounding error of the operand it emits (band) and, on the text side, scores the row exactly against its
 * ground-truth video -- the 235 MB that laff_rank_prepare reads back at 40k x 10k are never read.  Two launches, videos first:
 *   side 2 (video rows): band = band_v [((N + 3) & ~3) + ceil(N / 64)], per-column values written;
 *   side 1 (text rows) : band = band_t [N]; Ev / Nv / band_v = the video side's E, row count and band_v (its 64-column block maxima
 *                        are completed here: N * 16 >= Nv); s_gt64 [N] (-inf where gt_col - col0 is outside [0, Nv)), count [N]
 *                        cleared, the header of `pairs` cleared.
 * What the exact-rank pipeline needs afterwards is exactly what laff_rank_prepare leaves behind: laff_sim_gemm_banded and
 * laff_rank_resolve follow unchanged; s_gt64 has the arithmetic of laff_rank_resolve's re-score (equal rows give bit-equal scores).
 * LAFF_E_UNSUPPORTED for shapes this path does not cover (d > 512, unsplit heads, no operand): call laff_rank_prepare instead. */
typedef struct laff_rank_side {
    int side;
    const int* gt_col;
    int col0;
    const float* Ev;
    int Nv;
    double* s_gt64;
    float* band;
    float* band_v;
    int* count;
    unsigned* pairs;
    double* partials;      /* H > 1: scratch [N][H][2] (the per-head terms of a row come from several workgroups; the last arrival sums them */
    unsigned* tickets;     /*        in head order: the arithmetic of laff_rank_resolve's re-score); [N], cleared by the call */
} laff_rank_side;
int laff_fuse_packed_rank(laff_ctx* ctx, const laff_plane* planes, int L, int N, int H, int d, const float* w, const float* b,
                          const float* gw, unsigned flags, float* E, float* attn_w, void* E16, int precision, float prescale,
                          const laff_rank_side* rs /* NULL: plain laff_fuse_packed */);

/* ---- a7: per-video frame attention of VisMutiTransformNetPlusFrameFeat (model/model.py:2163-2173) --------
 * V[B,d] = Attention_1 over the Fmax frames of each video; frames[B,Fmax,d] zero padded.
 * lens != NULL: frames >= lens[b] are known zeros and are accounted for analytically (they still take
 * part in the softmax and in mean_L exactly as in the reference, whose mask slice is a no-op);
 * lens == NULL: all Fmax frames are read (needed behind vis_frame_addFC, where padding becomes the bias). */
int laff_frame_fuse(laff_ctx* ctx, const float* frames, const int* lens, int B, int Fmax, int d,
                    const float* w /*[d]*/, const float* b /*[1]*/, const float* gw /*[1]*/, unsigned flags,
                    float* V);
/* the frame features of one tower (same B / Fmax / d / flags / lens, own frames, attention parameters and output) in ONE
 * launch: host arrays of `count` device pointers (gw may be NULL without WITH_AVE) */
int laff_frame_fuse_grouped(laff_ctx* ctx, int count, const float* const* frames, const int* lens, int B, int Fmax, int d,
                            const float* const* w, const float* const* b, const float* const* gw, unsigned flags,
                            float* const* V);
/* the same, fed with the reference's mask_tensor (model/model.py:2156-2160: fp32 (B, ldm >= Fmax), ones for the frames a video has,
 * then zeros) instead of lens: lens[b] = round(sum_f mask[b][f]) is taken inside the launch -- `mask.sum(dim=1).int()` as two
 * framework launches in front of this one was 15 us of a 0.47 ms pass (C3) */
int laff_frame_fuse_grouped_mask(laff_ctx* ctx, int count, const float* const* frames, const float* mask, int ldm, int B, int Fmax,
                                 int d, const float* const* w, const float* const* b, const float* const* gw, unsigned flags,
                                 float* const* V);

/* ---- a8: loss.l2norm (loss.py:8-13) + operand packing for the similarity GEMM ---------------------------
 * For each (n,h): y = x / (||x||_2 + eps + 1e-14) if normalize, then y * prescale, converted to `precision`.
 * out layout: FP32 -> float [N, H*d]; FP16/BF16 -> 16-bit [N, H*d]; *X3 -> two planes [2][N, H*d] (hi, lo).
 * laff_packed_bytes gives the size. */
int laff_packed_bytes(int N, int K, int precision, size_t* out);
int laff_pack_rows(laff_ctx* ctx, const float* E, int N, int H, int d, int lde, int normalize, float eps,
                   float prescale, int precision, void* out);

/* ---- a9-a11: loss.cosine_sim + W2VVPP.get_txt2vis_matrix (loss.py:30-34, model/model.py:1003-1016) ------
 * S[Nt,Nv] = scale * T[Nt,K] . V[Nv,K]^T on operands produced by laff_pack_rows (K = H*d, scale = 1/(H*prescale^2)).
 * S may be NULL when only ranks are wanted.  If gt_col != NULL the epilogue also accumulates
 *   count[t] += #{ v : v + col0 != gt_col[t] and S[t,v] > s_gt[t] }   (count must be zeroed by the caller)
 * which is the argsort/label loop of predictor.py:232-244 in count form.  Any K (even for 16-bit
 * operands); K bytes a multiple of 128 takes the fast direct-to-LDS path, other sizes the generic staging paths. */
int laff_sim_gemm(laff_ctx* ctx, const void* T, const void* V, int Nt, int Nv, int K, float scale,
                  int precision, float* S, int lds, const int* gt_col, int col0, const float* s_gt,
                  int* count);

/* Ground-truth pre-pass for the fused count: s_gt[t] = scale * <T[t], V[gt_col[t]-col0]> on the packed operands
 * (fp32 accumulation of the same 16-bit products), -inf when that column is outside [0,Nv).  When laff_sim_gemm is
 * then called with gt_col/s_gt it writes exactly this value at S[t, gt] so counts and S stay consistent.
 * 16-bit precisions only.  zero_count (nullable, [Nt]) is cleared on the way: it is the accumulator the fused count of
 * laff_sim_gemm adds into (saves a fill launch). */
int laff_row_dot_gt(laff_ctx* ctx, const void* T, const void* V, int Nt, int Nv, int K, float scale, int precision,
                    const int* gt_col, int col0, float* s_gt, int* zero_count);

/* Host-side helper (no device work, any thread): owner[t] = position in the video id list of the prefix of caption id t before its first
 * '#' -- the match predictor.py:241 makes per query (`txt_id.split('#')[0]` looked up in vis_ids).  Both id lists arrive as one UTF-8 blob
 * each, ids separated by '\n', no trailing separator (ids must not contain '\n').  LAFF_E_SHAPE + message when a video id occurs twice
 * or a caption names a video that is not there. */
int laff_match_ids(const char* txt_blob, size_t txt_bytes, int n_txt, const char* vis_blob, size_t vis_bytes, int n_vis, int* owner /*host*/);

/* laff_rank_prepare for ONE side of a sharded pass (laff_amd/dist.py 'video16': the 16-bit text operand is what is all-gathered, the
 * fp32 text rows stay with their owner):
 *   sides = 1, text rows : s_gt64 [Nt] = the exact score of every text against Ev[gt_col - col0] (-inf outside [0, Nv)), band_t [Nt],
 *                          count cleared, list header cleared.  Ev = the fp32 video rows present (V, band_v untouched, may be NULL);
 *   sides = 2, video rows: band_v [((Nv + 3) & ~3) + ceil(Nv / 64)] (Et, T, gt_col, s_gt64, band_t untouched, may be NULL). */
int laff_rank_prepare_part(laff_ctx* ctx, int sides, const float* Et, const float* Ev, const void* T, const void* V, int Nt, int Nv, int H,
                           int d, int precision, float prescale, const int* gt_col, int col0, double* s_gt64, float* band_t, float* band_v,
                           int* zero_count, unsigned* pairs);

/* laff_rank_prepare that PRODUCES one or both GEMM operands on the way (emit: 1 = T, 2 = V, 3 = both): E16 = fp16 / bf16 (E * prescale),
 * no re-normalisation -- what laff_pack_rows(normalize = 0) writes for rows that are unit-norm already (the towers' outputs; gathered
 * embedding rows of a sharded pass, laff_amd/dist.py): one launch and one pass over those rows instead of two.  The operand not named in
 * `emit` is read as in laff_rank_prepare.  Single-plane 16-bit precisions only (LAFF_E_UNSUPPORTED otherwise). */
int laff_rank_prepare_emit(laff_ctx* ctx, int emit, const float* Et, const float* Ev, void* T, void* V, int Nt, int Nv, int H, int d,
                           int precision, float prescale, const int* gt_col, int col0, double* s_gt64, float* band_t, float* band_v,
                           int* zero_count, unsigned* pairs);

/* The listed pairs of a video shard's laff_sim_gemm_banded, exported to the owners of the TEXT rows instead of re-scored here (the exact
 * re-score needs the fp32 text row).  bounds [world + 1] (device): owner o holds text rows [bounds[o], bounds[o + 1]).  out: `world`
 * buckets of `cap` slots {row - bounds[o], col + col0}, unused slots 0xffffffff (the call fills them); fill [world + 1]: pairs per
 * bucket, [world] = 1 if a bucket overflowed (then count[0] is poisoned like an overflowing list).  Behind a 4-word header
 * {0, 0, n_slots, 4} any concatenation of buckets is a list laff_rank_resolve reads.  S (optional): the ground-truth entries take
 * the exact score, as laff_rank_resolve would write them.  predictor.py:232-244 is the loop this pipeline replaces. */
int laff_rank_export_pairs(laff_ctx* ctx, const double* s_gt64, int* count, float* S, int lds, int Nv, unsigned* pairs, unsigned pair_cap,
                           const int* bounds /*device*/, int world, int col0, unsigned* out, unsigned cap, unsigned* fill);

/* ---- a12 made EXACT on a reduced-precision GEMM ------------------------------------------------------------------------------
 * The reference ranks come from fp32 scores (predictor.py:232-244 on model/model.py:1003-1016).  A 16-bit MFMA pass keeps the
 * scores inside the 1e-4 contract but not the ranks, so the fused count is done against a PROVEN error band and the few pairs
 * inside the band are re-scored exactly.  exact(t, v) = (1/H) sum_h <t_h, v_h> / ((|t_h| + eps)(|v_h| + eps)), eps = 1e-13 + 1e-14
 * (loss.py:8-13, 30-34), evaluated in fp64 on the fp32 embeddings; rank = 1 + #{v != gt : exact(t, v) > exact(t, gt)}.
 *
 *   laff_rank_prepare      Et [Nt, H, d], Ev [Nv, H, d]: the fp32 embeddings (rows 16-byte aligned, d % 4 == 0); T, V: the GEMM
 *                          operands made from them (laff_pack_rows / laff_fuse_packed; `precision`, `prescale` as given there).
 *                          Writes s_gt64[t] = exact(t, gt_col[t] - col0) (-inf when that column is outside [0, Nv): another
 *                          shard owns it -- all-reduce MAX the doubles), band_t [Nt] and band_v with
 *                          |S_gemm(t, v) - exact(t, v)| <= band_t[t] + band_v[v]  (measured operand rounding error of both rows by
 *                          Cauchy-Schwarz + the fp32 accumulation bound; behind the per-column values, from offset (Nv + 3) & ~3:
 *                          the maximum of every aligned block of 64 columns, which is what the GEMM epilogue uses -- band_v holds
 *                          ((Nv + 3) & ~3) + ceil(Nv / 64) floats), clears zero_count [Nt] (nullable) and the pair-list header.
 *   laff_sim_gemm_banded   (gt_col, s_gt64, band_t, band_v: 16-byte aligned and readable up to the next multiple of 16 bytes -- the
 *                          kernel fetches them in 16-byte groups by LDS-DMA.)  laff_sim_gemm whose fused count is exact-decidable: count[t] += #{v != gt : S > s_gt + band}, pairs with
 *                          |S - s_gt| <= band are listed in `pairs` (uint32: header {n_extra, overflow, A, w} then pair_cap x
 *                          {row, col}, col shard-local: A slots in per-wavefront segments of w slots -- valid pairs first, unused
 *                          slots have row 0xffffffff -- then n_extra pairs appended by wavefronts whose segment was too small);
 *                          S (nullable) receives (float)s_gt64 at the ground-truth entry.
 *   laff_rank_resolve      re-scores the listed pairs: count[row] += exact > s_gt64[row]; S (nullable) takes the fp32 value of the
 *                          exact score (one ulp above (float)s_gt64 where rounding would hide a strict inequality) so that ranks
 *                          recounted from S equal count + 1.  pair_cap is used in whole groups of four slots (rounded down, >= 4).
 *                          More than pair_cap pairs: pairs[1] = 1 and count[0] is poisoned with -(2^26) -- negative even after an
 *                          int32 all-reduce SUM over <= 16 shards -- which trips the rank < 1 flag of laff_rank_metrics*.
 * After laff_rank_resolve (and an all-reduce SUM of count when videos are sharded) count + 1 are the exact ranks. */
int laff_rank_prepare(laff_ctx* ctx, const float* Et, const float* Ev, const void* T, const void* V, int Nt, int Nv, int H, int d,
                      int precision, float prescale, const int* gt_col, int col0, double* s_gt64, float* band_t, float* band_v,
                      int* zero_count, unsigned* pairs);
int laff_sim_gemm_banded(laff_ctx* ctx, const void* T, const void* V, int Nt, int Nv, int K, float scale, int precision, float* S,
                         int lds, const int* gt_col, int col0, const double* s_gt64, const float* band_t, const float* band_v,
                         int* count, unsigned* pairs, unsigned pair_cap);
int laff_rank_resolve(laff_ctx* ctx, const float* Et, const float* Ev, int Nt, int Nv, int H, int d, const double* s_gt64,
                      int* count, float* S, int lds, unsigned* pairs, unsigned pair_cap);
/* laff_rank_resolve and laff_rank_metrics[_async](count, base) in ONE launch, for passes in which nothing (no all-reduce of the counts)
 * comes between them: the workgroup of the resolve launch that finishes last turns the final counts into ranks (ranks_out, nullable:
 * count + base) and the seven metrics of evaluation.eval (/root/reference/evaluation.py:92-109) -- the label loop + eval of
 * predictor.py:232-246 end in the same kernel that settles the last rank.  out8 (host): 7 metrics + flag as laff_rank_metrics_async.
 *   synchronous == 0: returns at once; when out8 is pinned host memory (hipHostMalloc / hipHostRegister) the device writes it
 *                     directly -- valid after the stream (or the graph the call was captured in) has completed --, otherwise a 64-byte
 *                     copy is queued behind the launch.  Capturable (call it once eagerly first: the ctx allocates its result slots then).
 *   synchronous != 0: waits for the stream; LAFF_E_ARG when the flag is set (a rank < 1: overflowed pair list).
 * The fp64 mean of reciprocals is summed in this one workgroup's fixed order: equal to laff_rank_metrics to rounding (1e-16). */
int laff_rank_resolve_metrics(laff_ctx* ctx, const float* Et, const float* Ev, int Nt, int Nv, int H, int d, const double* s_gt64,
                              int* count, float* S, int lds, unsigned* pairs, unsigned pair_cap, int base, int* ranks_out,
                              double* out8 /*host*/, int synchronous);

/* s_gt[t] = S[t, gt_col[t]-col0] if that column is in [0,Nv) else -inf  (shard-local ground-truth score) */
int laff_gather_gt(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, const int* gt_col, int col0,
                   float* s_gt);
/* count[t] (+)= #{ v in [0,Nv) : v + col0 != gt_col[t] and S[t,v] > s_gt[t] }; accumulate != 0 adds. */
int laff_rank_count(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, const int* gt_col, int col0,
                    const float* s_gt, int* count, int accumulate);
/* Video-to-text direction (predictor.py:262-270): for every text t with owner video gt_col[t]:
 *   count[t] = #{ t' != t : S[t', v] > S[t, v] },  v = gt_col[t] - col0 in [0,Nv)  (others untouched).
 * grp_off[Nv+1] / grp_idx[Nt]: CSR of texts grouped by owner column (local). */
int laff_v2t_count(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, const int* grp_off,
                   const int* grp_idx, int max_group, int* count);

/* The same count made EXACT on the outputs of the exact-rank pipeline (replaces the column argsort of /root/reference/predictor.py:262-270
 * with positions that do not depend on the operand precision of the GEMM):
 *   count[t] = #{ t' != t : exact(t', v) > exact(t, v) },  exact() = the fp64 cosine of the fp32 embeddings (laff_rank_prepare).
 * S must be what laff_sim_gemm_banded (+ laff_rank_resolve) wrote for these operands: every entry within band_t[t'] + band_v[v] of
 * exact(t', v).  Entries further than that from a caption's threshold s_gt64[t] are decided on S; the others are listed as {t', v, t}
 * (3 words each behind a 4-word header {entries wanted, overflow flag, 0, 0}; list_cap entries) and re-scored in fp64 from Et / Ev.
 * count [Nt] is overwritten (0 for texts whose video is not a column of this S).  list[1] != 0 afterwards: the list was too small,
 * count is not valid -- call again with list_cap >= list[0]. */
int laff_v2t_count_exact(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, const int* grp_off, const int* grp_idx,
                         int max_group, const float* Et, const float* Ev, int H, int d, const double* s_gt64,
                         const float* band_t, const float* band_v, int* count, unsigned* list, unsigned list_cap);

/* ---- result lists (predictor.txt2video_write_to_file, predictor.py:53-88): for every row the K best columns, score
 * descending (ties: larger column first = a stable ascending argsort read backwards), instead of a full-matrix argsort.
 * idx_out [Nt,K] int32, val_out [Nt,K] fp32.  1 <= K <= min(Nv, 8192); the row (4 Nv bytes) + 8 K' bytes (K' = K rounded up to
 * 64 / 512 / 2048 / 4096 / 8192) must fit 160 KiB of LDS: wider collections are split by columns and the per-block lists merged with a
 * second call (laff_amd.ops.topk_rows does; LAFF_E_UNSUPPORTED says how many columns fit). */
int laff_topk_rows(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, int K, int* idx_out, float* val_out);

/* ---- a13: evaluation.eval (evaluation.py:92-109) for single-GT rows ---------------------------------------
 * rank[i] = r[i] + base must be >= 1: pass 1-based ranks with base = 0, or the counts of better-scoring videos that
 * laff_sim_gemm / laff_rank_count produce with base = 1; ranks_out (nullable, [Nq] device) receives the ranks.
 * out7 (host) = r1, r5, r10, medr, meanr, mir, mAP.  Reduced on the device (one small kernel), 56 bytes copied back;
 * synchronises the stream. */
int laff_rank_metrics(laff_ctx* ctx, const int* r, int Nq, int base, int* ranks_out, double out7[7]);
/* Same, without synchronising: out8 is PINNED HOST memory (8 doubles: the 7 metrics + an error flag, 1.0 if a rank < 1 was
 * seen -- the 7 metrics are then NaN -- else 0.0); valid once the stream has been synchronised.  Capturable in a HIP graph.
 * The device writes a pinned (device-addressable) buffer directly from the kernel; any other host pointer gets a 64-byte copy queued
 * behind the launch. */
int laff_rank_metrics_async(laff_ctx* ctx, const int* r, int Nq, int base, int* ranks_out, double* out8_pinned_host);

/* ---- e: the collectives of the sharded path for a host that is not Python (laff_amd/dist.py issues the same three through
 * torch.distributed).  One process per GPU; RCCL is looked up at the first call (a copy already loaded into
 * the process first, then librccl.so / librccl.so.1) -- the library does not link against it, LAFF_E_UNSUPPORTED when there is none.
 * Every call is asynchronous on the stream the comm was created with (the ctx's; laff_comm_set_stream re-binds it).
 *   'video' scheme (the reference's loop model/model.py:1057-1077 sharded by video rows): laff_allgather_rows on the text
 *   embeddings, laff_allreduce_f64_max on s_gt64 (laff_rank_prepare writes -inf for texts whose video lives elsewhere),
 *   laff_allreduce_i32_sum on the counts (the overflow poison -2^26 survives a sum over up to 16 shards). */
#define LAFF_COMM_ID_BYTES 128
typedef struct laff_comm laff_comm;
int laff_comm_unique_id(unsigned char* id /*[LAFF_COMM_ID_BYTES], made by one rank, handed to the others out of band*/);
int laff_comm_init(laff_ctx* ctx, int rank, int world, const unsigned char* id, laff_comm** out);
int laff_comm_set_stream(laff_comm* comm, void* hip_stream);
int laff_comm_destroy(laff_comm* comm);
int laff_allgather_rows(laff_comm* comm, const void* send, void* recv, size_t bytes_per_rank);     /* recv: world x bytes, rank order */
int laff_allreduce_i32_sum(laff_comm* comm, int* buf, size_t n);                                    /* in place */
int laff_allreduce_f64_max(laff_comm* comm, double* buf, size_t n);                                 /* in place */

#ifdef __cplusplus
}
#endif
#endif /* LAFF_HIP_H */
