// api.hip -- extern "C" surface of liblaff_hip.so (see include/laff_hip.h for the contract).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

constexpr int METRIC_SLOTS = 33;

struct laff_ctx {
    int device;
    hipStream_t stream;
    double* d_metrics = nullptr;   // METRIC_SLOTS x (7 doubles + 1 flag: 0.0 / 1.0 = a rank < 1 was seen, metrics are NaN) on the device
    unsigned metrics_slot = 0;     // every laff_rank_metrics_async call takes the next slot: calls captured into different graphs (or in
                                   // flight on different streams) do not share their result buffer
    double* h_metrics = nullptr;   // pinned host mirror
    char* d_mscratch = nullptr;    // METRIC_SLOTS x rank_metrics_scratch_bytes(): the metrics kernel's partials + histograms, kept zero
};

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(LAFF_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define CHECK_CTX(ctx) \
    if (!(ctx)) return fail(LAFF_E_ARG, "%s: null ctx", __func__)
#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

int metrics_buffers(laff_ctx* ctx) {
    if (ctx->d_metrics) return LAFF_OK;
    const size_t sb = laff::rank_metrics_scratch_bytes();
    HIP_TRY(hipMalloc((void**)&ctx->d_mscratch, METRIC_SLOTS * sb));
    HIP_TRY(hipMemset(ctx->d_mscratch, 0, METRIC_SLOTS * sb));
    HIP_TRY(hipHostMalloc((void**)&ctx->h_metrics, 8 * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&ctx->d_metrics, METRIC_SLOTS * 8 * sizeof(double)));
    HIP_TRY(hipMemset(ctx->d_metrics, 0, METRIC_SLOTS * 8 * sizeof(double)));
    return LAFF_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

bool is_x3(int p) { return p == LAFF_PREC_FP16X3 || p == LAFF_PREC_BF16X3; }
int elem_size(int p) { return p == LAFF_PREC_FP32 ? 4 : 2; }

}  // namespace

extern "C" {

int laff_abi_version(void) { return LAFF_ABI_VERSION; }

const char* laff_last_error(void) { return g_err.c_str(); }

int laff_ctx_create(int device, void* hip_stream, laff_ctx** out) {
    if (!out) return fail(LAFF_E_ARG, "laff_ctx_create: null out");
    // tuning knobs: process-wide, re-read whenever a ctx is created; an absent variable means the default (not "the last value": a
    // test that set LAFF_STRIP=2, dropped it and made a new ctx used to leave every later bf16 similarity on the strip kernel)
    auto knob = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    laff::g_gemm_variant = knob("LAFF_GEMM_VARIANT", 0);
    laff::g_strip_mode = knob("LAFF_STRIP", 1);
    laff::g_strip_map = knob("LAFF_STRIP_MAP", 1);
    laff::g_fc_strip = knob("LAFF_FC_STRIP", 1);
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(LAFF_E_ARG, "laff_ctx_create: device %d out of range (%d devices)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LAFF_E_UNSUPPORTED, "laff_ctx_create: device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);
    if (prop.multiProcessorCount > 0) laff::g_num_cus = prop.multiProcessorCount;
    laff_ctx* c = new laff_ctx();
    c->device = device;
    c->stream = static_cast<hipStream_t>(hip_stream);
    *out = c;
    return LAFF_OK;
}

/* internal (comm.hip): the ctx's stream and device; the error text of the calling thread */
int laff_ctx_stream_device(laff_ctx* ctx, void** stream, int* device) {
    CHECK_CTX(ctx);
    *stream = (void*)ctx->stream;
    *device = ctx->device;
    return LAFF_OK;
}
void laff_set_error(const char* msg) { g_err = msg ? msg : ""; }

int laff_ctx_set_stream(laff_ctx* ctx, void* hip_stream) {
    CHECK_CTX(ctx);
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return LAFF_OK;
}

int laff_ctx_destroy(laff_ctx* ctx) {
    if (ctx) {
        if (ctx->d_metrics) (void)hipFree(ctx->d_metrics);
        if (ctx->d_mscratch) (void)hipFree(ctx->d_mscratch);
        if (ctx->h_metrics) (void)hipHostFree(ctx->h_metrics);
    }
    delete ctx;
    return LAFF_OK;
}

namespace {
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
}  // namespace

int laff_stamp(laff_ctx* ctx, unsigned long long* slot) {
    CHECK_CTX(ctx);
    if (!slot) return fail(LAFF_E_ARG, "laff_stamp: null slot");
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ctx->stream, slot);
    HIP_TRY(hipGetLastError());
    return LAFF_OK;
}

int laff_wall_clock_khz(laff_ctx* ctx, int* khz) {
    CHECK_CTX(ctx);
    if (!khz) return fail(LAFF_E_ARG, "laff_wall_clock_khz: null out");
    HIP_TRY(hipDeviceGetAttribute(khz, hipDeviceAttributeWallClockRate, ctx->device));
    return LAFF_OK;
}

int laff_device_info(laff_ctx* ctx, int out[4]) {
    CHECK_CTX(ctx);
    if (!out) return fail(LAFF_E_ARG, "laff_device_info: null out");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    out[0] = prop.multiProcessorCount;
    out[1] = prop.clockRate / 1000;
    out[2] = (int)prop.sharedMemPerBlock;
    out[3] = prop.warpSize;
    return LAFF_OK;
}

static int fc_problem_args(const laff_fc_problem& q, laff::GemmArgs& a, bool& glds, const char* who) {
    if (q.N == 0) { a = laff::GemmArgs{}; glds = true; return LAFF_OK; }      /* empty problem (skipped by the callers) */
    if (!q.X || !q.W || !q.Y) return fail(LAFF_E_ARG, "%s: null X/W/Y", who);
    if (q.N < 0 || q.Dk < 1 || q.D < 1 || q.ldx < q.Dk || q.ldw < q.Dk || q.ldy < q.D)
        return fail(LAFF_E_SHAPE, "%s: bad shape N=%d Dk=%d D=%d ldx=%d ldw=%d ldy=%d", who, q.N, q.Dk, q.D, q.ldx, q.ldw, q.ldy);
    if (q.act < LAFF_ACT_NONE || q.act > LAFF_ACT_SIGMOID) return fail(LAFF_E_ARG, "%s: bad act %d", who, q.act);
    if ((q.bn_scale == nullptr) != (q.bn_shift == nullptr)) return fail(LAFF_E_ARG, "%s: bn_scale/bn_shift must come together", who);
    if ((q.bias && !aligned16(q.bias)) || (q.bn_scale && (!aligned16(q.bn_scale) || !aligned16(q.bn_shift))))
        return fail(LAFF_E_ALIGN, "%s: bias / bn_scale / bn_shift must be 16-byte aligned", who);
    a = laff::GemmArgs{};
    a.R = q.X; a.C = q.W; a.nR = q.N; a.nC = q.D; a.K = q.Dk; a.ldR = q.ldx; a.ldC = q.ldw;
    a.nseg = 1; a.segR[0] = a.segC[0] = 0;
    a.out = q.Y; a.ldo = q.ldy; a.scale = 1.0f;
    a.bias = q.bias; a.bn_scale = q.bn_scale; a.bn_shift = q.bn_shift; a.act = q.act;
    glds = aligned16(q.X) && aligned16(q.W) && (q.ldx % 4 == 0) && (q.ldw % 4 == 0) && (q.Dk % 4 == 0);
    return LAFF_OK;
}

int laff_fc_act_bn_grouped(laff_ctx* ctx, const laff_fc_problem* problems, int count);

int laff_fc_act_bn(laff_ctx* ctx, const float* X, int N, int Dk, int ldx, const float* W, int ldw, const float* bias,
                   const float* bn_scale, const float* bn_shift, int D, int act, float* Y, int ldy) {
    CHECK_CTX(ctx);
    laff_fc_problem q{X, N, Dk, ldx, W, ldw, bias, bn_scale, bn_shift, D, act, Y, ldy};
    return laff_fc_act_bn_grouped(ctx, &q, 1);
}

int laff_fc_act_bn_grouped(laff_ctx* ctx, const laff_fc_problem* problems, int count) {
    CHECK_CTX(ctx);
    if (!problems || count < 0) return fail(LAFF_E_ARG, "laff_fc_act_bn_grouped: bad problem list");
    DeviceGuard g(ctx->device);
    // problems are grouped by staging kind (see staging_kind), at most MAX_GROUP per launch
    for (int kind = 2; kind >= 0; --kind) {
        laff::GroupedGemmArgs ga{};
        for (int i = 0; i < count; ++i) {
            laff::GemmArgs a;
            bool aligned;
            if (int rc = fc_problem_args(problems[i], a, aligned, "laff_fc_act_bn_grouped")) return rc;
            if (problems[i].N == 0 || laff::staging_kind(a, 4, aligned) != kind) continue;
            ga.p[ga.count++] = a;
            if (ga.count == laff::MAX_GROUP) {
                HIP_TRY(laff::launch_gemm_nt_grouped_f32(ga, kind, ctx->stream));
                ga.count = 0;
            }
        }
        if (ga.count) HIP_TRY(laff::launch_gemm_nt_grouped_f32(ga, kind, ctx->stream));
    }
    return LAFF_OK;
}

int laff_fc_gather_act_bn(laff_ctx* ctx, const int* indptr, const int* indices, const float* values, int N, int Dk,
                          const float* Wt, int ldwt, const float* bias, const float* bn_scale, const float* bn_shift, int D,
                          int act, float* Y, int ldy) {
    CHECK_CTX(ctx);
    if (N == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!indptr || !indices || !Wt || !Y) return fail(LAFF_E_ARG, "laff_fc_gather_act_bn: null argument");
    if (N < 0 || Dk < 1 || D < 4 || (D & 3) || D > 8192 || ldwt < D || (ldwt & 3) || ldy < D || (ldy & 3))
        return fail(LAFF_E_SHAPE, "laff_fc_gather_act_bn: bad shape N=%d Dk=%d D=%d ldwt=%d ldy=%d", N, Dk, D, ldwt, ldy);
    if (act < LAFF_ACT_NONE || act > LAFF_ACT_SIGMOID) return fail(LAFF_E_ARG, "laff_fc_gather_act_bn: bad act %d", act);
    if ((bn_scale == nullptr) != (bn_shift == nullptr)) return fail(LAFF_E_ARG, "laff_fc_gather_act_bn: bn_scale/bn_shift must come together");
    if (!aligned16(Wt) || !aligned16(Y)) return fail(LAFF_E_ALIGN, "laff_fc_gather_act_bn: Wt / Y must be 16-byte aligned");
    if (N == 0) return LAFF_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_fc_gather(indptr, indices, values, N, Dk, Wt, ldwt, bias, bn_scale, bn_shift, D, act, Y, ldy, ctx->stream));
    return LAFF_OK;
}

namespace {
struct LossLayout {
    size_t XH, XHT, nrm, npr, S, dS, dST, G, loss_h, total;   // offsets in floats
    int dp, Bp;
};
LossLayout loss_layout(int B, int H, int d) {
    LossLayout L{};
    L.dp = (d + 3) & ~3;
    L.Bp = (B + 3) & ~3;
    auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    size_t o = 0;
    L.XH = o;     o += up4((size_t)2 * H * B * L.dp);
    L.XHT = o;    o += up4((size_t)2 * H * d * L.Bp);
    L.nrm = o;    o += up4((size_t)2 * H * B);
    L.npr = o;    o += up4((size_t)2 * H * B);
    L.S = o;      o += up4((size_t)H * B * L.Bp);
    L.dS = o;     o += up4((size_t)H * B * L.Bp);
    L.dST = o;    o += up4((size_t)H * B * L.Bp);
    L.G = o;      o += up4((size_t)2 * H * B * L.dp);
    L.loss_h = o; o += up4((size_t)H);
    L.total = o;
    return L;
}
}  // namespace

int laff_margin_loss_workspace_bytes(int B, int H, int d, size_t* out) {
    if (!out || B < 1 || H < 1 || d < 1) return fail(LAFF_E_ARG, "laff_margin_loss_workspace_bytes: bad args");
    *out = loss_layout(B, H, d).total * sizeof(float);
    return LAFF_OK;
}

int laff_margin_loss(laff_ctx* ctx, const float* s, const float* im, int B, int H, int d, float margin, unsigned flags,
                     float* loss, float* d_s, float* d_im, void* workspace, size_t workspace_bytes) {
    CHECK_CTX(ctx);
    if (!s || !im || !loss || !workspace) return fail(LAFF_E_ARG, "laff_margin_loss: null argument");
    if (B < 1 || H < 1 || d < 1 || B > 16384) return fail(LAFF_E_SHAPE, "laff_margin_loss: bad shape B=%d H=%d d=%d", B, H, d);
    if (flags & ~15u) return fail(LAFF_E_ARG, "laff_margin_loss: unknown flags 0x%x", flags);
    const LossLayout L = loss_layout(B, H, d);
    if (workspace_bytes < L.total * sizeof(float)) return fail(LAFF_E_ARG, "laff_margin_loss: workspace too small (%zu < %zu bytes)", workspace_bytes, L.total * sizeof(float));
    if (!aligned16(workspace)) return fail(LAFF_E_ALIGN, "laff_margin_loss: workspace must be 16-byte aligned");
    if ((size_t)(2 * B + 16) * sizeof(float) > 64 * 1024) return fail(LAFF_E_UNSUPPORTED, "laff_margin_loss: B=%d exceeds the reduction kernel's LDS budget", B);
    DeviceGuard g(ctx->device);
    float* ws = (float*)workspace;
    const int dp = L.dp, Bp = L.Bp;
    const int use_s = (flags & LAFF_LOSS_DIR_I2T) ? 1 : 0, use_im = (flags & LAFF_LOSS_DIR_T2I) ? 1 : 0;
    const int maxv = (flags & LAFF_LOSS_MAX_VIOLATION) ? 1 : 0;
    const float gmean = maxv ? 1.0f / (float)B : 1.0f / ((float)B * (float)B);
    const float gw = (flags & LAFF_LOSS_COST_MEAN) ? gmean : 1.0f;
    HIP_TRY(laff::launch_loss_normalize(s, im, B, H, d, dp, Bp, 1e-13f, ws + L.XH, ws + L.XHT, ws + L.nrm, ws + L.npr, ctx->stream));
    auto XH = [&](int z, int h) { return ws + L.XH + ((size_t)z * H + h) * B * dp; };
    auto XHT = [&](int z, int h) { return ws + L.XHT + ((size_t)z * H + h) * d * Bp; };
    auto G = [&](int z, int h) { return ws + L.G + ((size_t)z * H + h) * B * dp; };
    std::vector<laff_fc_problem> probs((size_t)H);
    for (int h = 0; h < H; ++h)      // scores_h [B videos, B captions] = I^_h . S^_h^T
        probs[h] = laff_fc_problem{XH(1, h), B, d, dp, XH(0, h), dp, nullptr, nullptr, nullptr, B, LAFF_ACT_NONE,
                                   ws + L.S + (size_t)h * B * Bp, Bp};
    if (int rc = laff_fc_act_bn_grouped(ctx, probs.data(), H)) return rc;
    HIP_TRY(laff::launch_margin_reduce(ws + L.S, ws + L.dS, ws + L.dST, ws + L.loss_h, loss, B, Bp, H, margin, maxv, use_s, use_im,
                                       gw, gw, ctx->stream));
    if (!d_s && !d_im) return LAFF_OK;
    probs.clear();
    for (int h = 0; h < H; ++h) {
        // dL/dI^_h = dS_h . S^_h   (column operand = S^_h^T, K = captions);  dL/dS^_h = dS_h^T . I^_h
        probs.push_back(laff_fc_problem{ws + L.dS + (size_t)h * B * Bp, B, B, Bp, XHT(0, h), Bp, nullptr, nullptr, nullptr, d,
                                        LAFF_ACT_NONE, G(1, h), dp});
        probs.push_back(laff_fc_problem{ws + L.dST + (size_t)h * B * Bp, B, B, Bp, XHT(1, h), Bp, nullptr, nullptr, nullptr, d,
                                        LAFF_ACT_NONE, G(0, h), dp});
    }
    if (int rc = laff_fc_act_bn_grouped(ctx, probs.data(), (int)probs.size())) return rc;
    HIP_TRY(laff::launch_loss_normalize_bwd(ws + L.XH, ws + L.G, ws + L.nrm, ws + L.npr, B, H, d, dp, d_s, d_im, ctx->stream));
    return LAFF_OK;
}

int laff_split_rows_bytes(int N, int K, size_t* out) {
    if (!out || N < 0 || K < 1) return fail(LAFF_E_ARG, "laff_split_rows_bytes: bad args");
    const size_t Kp = (size_t)(K + 63) / 64 * 64;
    *out = 2 * (size_t)N * Kp * 2;
    return LAFF_OK;
}

int laff_split_rows(laff_ctx* ctx, const float* X, int N, int K, int ldx, void* out, float* rscale) {
    CHECK_CTX(ctx);
    if (N == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!X || !out || !rscale) return fail(LAFF_E_ARG, "laff_split_rows: null argument");
    if (N < 0 || K < 1 || ldx < K) return fail(LAFF_E_SHAPE, "laff_split_rows: bad shape N=%d K=%d ldx=%d", N, K, ldx);
    if (!aligned16(out)) return fail(LAFF_E_ALIGN, "laff_split_rows: out must be 16-byte aligned");
    if (N == 0) return LAFF_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_split_rows(X, N, K, ldx, (K + 63) / 64 * 64, out, rscale, ctx->stream));
    return LAFF_OK;
}

int laff_split_rows_grouped(laff_ctx* ctx, int count, const float* const* X, const int* N, const int* K, const int* ldx,
                            void* const* out, float* const* rscale) {
    CHECK_CTX(ctx);
    if (count < 0 || (count && (!X || !N || !K || !ldx || !out || !rscale))) return fail(LAFF_E_ARG, "laff_split_rows_grouped: bad argument list");
    DeviceGuard g(ctx->device);
    for (int i0 = 0; i0 < count; i0 += 8) {
        const int c = count - i0 < 8 ? count - i0 : 8;
        for (int i = i0; i < i0 + c; ++i) {
            if (!X[i] || !out[i] || !rscale[i]) return fail(LAFF_E_ARG, "laff_split_rows_grouped: matrix %d has a null pointer", i);
            if (N[i] < 0 || K[i] < 1 || ldx[i] < K[i]) return fail(LAFF_E_SHAPE, "laff_split_rows_grouped: matrix %d bad shape", i);
            if (!aligned16(out[i])) return fail(LAFF_E_ALIGN, "laff_split_rows_grouped: out %d must be 16-byte aligned", i);
        }
        HIP_TRY(laff::launch_split_rows_grouped(c, X + i0, N + i0, K + i0, ldx + i0, out + i0, rscale + i0, ctx->stream));
    }
    return LAFF_OK;
}

int laff_fc_act_bn_split_grouped(laff_ctx* ctx, const laff_fc_split_problem* problems, int count) {
    CHECK_CTX(ctx);
    if (!problems || count < 0) return fail(LAFF_E_ARG, "laff_fc_act_bn_split_grouped: bad problem list");
    DeviceGuard g(ctx->device);
    laff::GroupedGemmArgs ga{};
    for (int i = 0; i < count; ++i) {
        const laff_fc_split_problem& q = problems[i];
        if (!q.Xs || !q.Ws || !q.x_rscale || !q.w_rscale || !q.Y) return fail(LAFF_E_ARG, "laff_fc_act_bn_split_grouped: problem %d has a null operand", i);
        if (q.N < 0 || q.Dk < 1 || q.D < 1 || q.ldy < q.D) return fail(LAFF_E_SHAPE, "laff_fc_act_bn_split_grouped: problem %d bad shape", i);
        if (q.act < LAFF_ACT_NONE || q.act > LAFF_ACT_SIGMOID) return fail(LAFF_E_ARG, "laff_fc_act_bn_split_grouped: bad act %d", q.act);
        if ((q.bn_scale == nullptr) != (q.bn_shift == nullptr)) return fail(LAFF_E_ARG, "laff_fc_act_bn_split_grouped: bn_scale/bn_shift must come together");
        if (!aligned16(q.Xs) || !aligned16(q.Ws) || (q.bias && !aligned16(q.bias)) || (q.bn_scale && (!aligned16(q.bn_scale) || !aligned16(q.bn_shift))))
            return fail(LAFF_E_ALIGN, "laff_fc_act_bn_split_grouped: problem %d: 16-byte alignment", i);
        const int Kp = (q.Dk + 63) / 64 * 64;
        if ((long long)q.N * Kp * 4 >= (1ll << 32) || (long long)q.D * Kp * 4 >= (1ll << 32))
            return fail(LAFF_E_UNSUPPORTED, "laff_fc_act_bn_split_grouped: problem %d: packed operand exceeds 4 GiB", i);
        if (q.N == 0) continue;
        laff::GemmArgs a{};
        a.R = q.Xs; a.C = q.Ws; a.nR = q.N; a.nC = q.D; a.K = Kp; a.ldR = Kp; a.ldC = Kp;
        a.nseg = 3;                                   // lo*hi, hi*lo (small terms first), hi*hi
        a.segR[0] = (long)q.N * Kp * 2; a.segC[0] = 0;
        a.segR[1] = 0;                  a.segC[1] = (long)q.D * Kp * 2;
        a.segR[2] = 0;                  a.segC[2] = 0;
        a.out = q.Y; a.ldo = q.ldy; a.scale = 1.0f;
        a.row_scale = q.x_rscale; a.col_scale = q.w_rscale;
        a.bias = q.bias; a.bn_scale = q.bn_scale; a.bn_shift = q.bn_shift; a.act = q.act;
        ga.p[ga.count++] = a;
        if (ga.count == laff::MAX_GROUP) {
            HIP_TRY(laff::launch_gemm_nt_grouped_f16(ga, ctx->stream));
            ga.count = 0;
        }
    }
    if (ga.count) HIP_TRY(laff::launch_gemm_nt_grouped_f16(ga, ctx->stream));
    return LAFF_OK;
}

int laff_row_scales_grouped(laff_ctx* ctx, int count, const float* const* X, const int* N, const int* K, const int* ldx,
                            float* const* rscale) {
    CHECK_CTX(ctx);
    if (count < 0 || (count && (!X || !N || !K || !ldx || !rscale))) return fail(LAFF_E_ARG, "laff_row_scales_grouped: bad argument list");
    DeviceGuard g(ctx->device);
    for (int i0 = 0; i0 < count; i0 += 8) {
        const int c = count - i0 < 8 ? count - i0 : 8;
        void* none[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int i = i0; i < i0 + c; ++i) {
            if (N[i] == 0) continue;
            if (!X[i] || !rscale[i]) return fail(LAFF_E_ARG, "laff_row_scales_grouped: matrix %d has a null pointer", i);
            if (N[i] < 0 || K[i] < 1 || ldx[i] < K[i]) return fail(LAFF_E_SHAPE, "laff_row_scales_grouped: matrix %d bad shape", i);
        }
        HIP_TRY(laff::launch_split_rows_grouped(c, X + i0, N + i0, K + i0, ldx + i0, none, rscale + i0, ctx->stream));
    }
    return LAFF_OK;
}

int laff_fc_act_bn_fused_grouped(laff_ctx* ctx, const laff_fc_fused_problem* problems, int count) {
    CHECK_CTX(ctx);
    if (!problems || count < 0) return fail(LAFF_E_ARG, "laff_fc_act_bn_fused_grouped: bad problem list");
    DeviceGuard g(ctx->device);
    laff::GroupedGemmArgs ga{};
    for (int i = 0; i < count; ++i) {
        const laff_fc_fused_problem& q = problems[i];
        if (q.N == 0) continue;
        if (!q.X || !q.Ws || !q.x_rscale || !q.w_rscale || !q.Y) return fail(LAFF_E_ARG, "laff_fc_act_bn_fused_grouped: problem %d has a null operand", i);
        if (q.N < 0 || q.Dk < 32 || (q.Dk & 31) || q.D < 1 || q.ldy < q.D || q.ldx < q.Dk || (q.ldx & 3))
            return fail(LAFF_E_SHAPE, "laff_fc_act_bn_fused_grouped: problem %d: need Dk %% 32 == 0, ldx %% 4 == 0 (N=%d Dk=%d D=%d ldx=%d ldy=%d)",
                        i, q.N, q.Dk, q.D, q.ldx, q.ldy);
        if (q.act < LAFF_ACT_NONE || q.act > LAFF_ACT_SIGMOID) return fail(LAFF_E_ARG, "laff_fc_act_bn_fused_grouped: bad act %d", q.act);
        if ((q.bn_scale == nullptr) != (q.bn_shift == nullptr)) return fail(LAFF_E_ARG, "laff_fc_act_bn_fused_grouped: bn_scale/bn_shift must come together");
        if (!aligned16(q.X) || !aligned16(q.Ws) || (q.bias && !aligned16(q.bias)) || (q.bn_scale && (!aligned16(q.bn_scale) || !aligned16(q.bn_shift))))
            return fail(LAFF_E_ALIGN, "laff_fc_act_bn_fused_grouped: problem %d: 16-byte alignment", i);
        const int Kp = (q.Dk + 63) / 64 * 64;
        if ((long long)q.N * q.ldx * 4 >= (1ll << 32) || (long long)q.D * Kp * 4 >= (1ll << 32))
            return fail(LAFF_E_UNSUPPORTED, "laff_fc_act_bn_fused_grouped: problem %d: operand exceeds 4 GiB", i);
        laff::GemmArgs a{};
        a.Rf = q.X; a.ldRf = q.ldx; a.R = nullptr; a.ldR = 0;
        a.C = q.Ws; a.nR = q.N; a.nC = q.D; a.K = q.Dk; a.ldC = Kp;
        a.nseg = 3;
        a.segC[0] = 0; a.segC[1] = (long)q.D * Kp * 2; a.segC[2] = 0;
        a.out = q.Y; a.ldo = q.ldy; a.scale = 1.0f;
        a.row_scale = q.x_rscale; a.col_scale = q.w_rscale;
        a.bias = q.bias; a.bn_scale = q.bn_scale; a.bn_shift = q.bn_shift; a.act = q.act;
#ifdef LAFF_GEMM_TRACE
        if (const char* e = getenv("LAFF_GEMM_TRACE_PTR")) a.trace = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
        ga.p[ga.count++] = a;
        if (ga.count == laff::MAX_GROUP) {
            HIP_TRY(laff::launch_gemm_nt_x3_fused_grouped(ga, ctx->stream));
            ga.count = 0;
        }
    }
    if (ga.count) HIP_TRY(laff::launch_gemm_nt_x3_fused_grouped(ga, ctx->stream));
    return LAFF_OK;
}

/* host-side helper, no device work: owner[t] = position in the video id list of the prefix of caption id t before its first '#'
 * (predictor.py:241: txt_id.split('#')[0] looked up in vis_ids).  One open-addressing table over the video ids, FNV-1a. */
int laff_match_ids(const char* txt_blob, size_t txt_bytes, int n_txt, const char* vis_blob, size_t vis_bytes, int n_vis, int* owner) {
    if (n_txt < 0 || n_vis < 0 || (n_txt && (!txt_blob || !owner)) || (n_vis && !vis_blob)) return fail(LAFF_E_ARG, "laff_match_ids: bad arguments");
    if (n_txt == 0) return LAFF_OK;
    auto hash = [](const char* p, size_t n) {
        unsigned long long h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
        return h;
    };
    size_t cap = 16;
    while (cap < 2 * (size_t)n_vis + 2) cap <<= 1;
    struct Slot { const char* p; unsigned len; int idx; };
    std::vector<Slot> table(cap, Slot{nullptr, 0u, -1});
    size_t at = 0;
    for (int v = 0; v < n_vis; ++v) {
        if (at > vis_bytes) return fail(LAFF_E_ARG, "laff_match_ids: the video blob holds fewer than %d ids", n_vis);
        const char* b = vis_blob + at;
        const char* e = (const char*)memchr(b, '\n', vis_bytes - at);
        const size_t len = e ? (size_t)(e - b) : vis_bytes - at;
        at += len + 1;
        size_t s = hash(b, len) & (cap - 1);
        while (table[s].idx >= 0) {
            if (table[s].len == len && memcmp(table[s].p, b, len) == 0)
                return fail(LAFF_E_SHAPE, "laff_match_ids: video id '%.*s' appears twice in vis_ids", (int)len, b);
            s = (s + 1) & (cap - 1);
        }
        table[s] = Slot{b, (unsigned)len, v};
    }
    if (at != vis_bytes + 1 && !(n_vis == 0 && vis_bytes == 0)) return fail(LAFF_E_ARG, "laff_match_ids: the video blob does not hold exactly %d ids", n_vis);
    at = 0;
    for (int t = 0; t < n_txt; ++t) {
        if (at > txt_bytes) return fail(LAFF_E_ARG, "laff_match_ids: the caption blob holds fewer than %d ids", n_txt);
        const char* b = txt_blob + at;
        const char* e = (const char*)memchr(b, '\n', txt_bytes - at);
        const size_t line = e ? (size_t)(e - b) : txt_bytes - at;
        at += line + 1;
        const char* h = (const char*)memchr(b, '#', line);
        const size_t len = h ? (size_t)(h - b) : line;
        size_t s = hash(b, len) & (cap - 1);
        int found = -1;
        while (table[s].idx >= 0) {
            if (table[s].len == len && memcmp(table[s].p, b, len) == 0) { found = table[s].idx; break; }
            s = (s + 1) & (cap - 1);
        }
        if (found < 0) return fail(LAFF_E_SHAPE, "laff_match_ids: caption %d refers to a video that is not in vis_ids: '%.*s'", t, (int)len, b);
        owner[t] = found;
    }
    if (at != txt_bytes + 1) return fail(LAFF_E_ARG, "laff_match_ids: the caption blob does not hold exactly %d ids", n_txt);
    return LAFF_OK;
}

int laff_fc_strip_pack_bytes(int D, int Dk, size_t* out) {
    if (!out || D < 32 || (D & 31)) return fail(LAFF_E_SHAPE, "laff_fc_strip_pack_bytes: D must be a positive multiple of 32 (D=%d)", D);
    if (Dk != laff::FC_STRIP_K) return fail(LAFF_E_SHAPE, "laff_fc_strip_pack_bytes: the strip form takes Dk == 512 (Dk=%d)", Dk);
    *out = laff::fc_strip_image_bytes(D);
    return LAFF_OK;
}

int laff_fc_strip_pack(laff_ctx* ctx, const float* W, int ldw, const float* bias, const float* bn_scale, const float* bn_shift, int D,
                       int Dk, int act, void* img) {
    CHECK_CTX(ctx);
    if (!W || !img) return fail(LAFF_E_ARG, "laff_fc_strip_pack: null W / img");
    if (D < 32 || (D & 31) || Dk != laff::FC_STRIP_K || ldw < Dk)
        return fail(LAFF_E_SHAPE, "laff_fc_strip_pack: need D %% 32 == 0, Dk == 512, ldw >= Dk (D=%d Dk=%d ldw=%d)", D, Dk, ldw);
    if ((long long)D * 2048 >= (1ll << 32)) return fail(LAFF_E_UNSUPPORTED, "laff_fc_strip_pack: image exceeds 4 GiB");
    if (act < LAFF_ACT_NONE || act > LAFF_ACT_SIGMOID) return fail(LAFF_E_ARG, "laff_fc_strip_pack: bad act %d", act);
    if ((bn_scale == nullptr) != (bn_shift == nullptr)) return fail(LAFF_E_ARG, "laff_fc_strip_pack: bn_scale/bn_shift must come together");
    if (!aligned16(img)) return fail(LAFF_E_ALIGN, "laff_fc_strip_pack: img must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_fc_strip_pack(W, ldw, bias, bn_scale, bn_shift, D, act, img, ctx->stream));
    return LAFF_OK;
}

int laff_fc_act_bn_strip_grouped(laff_ctx* ctx, const laff_fc_strip_problem* problems, int count) {
    CHECK_CTX(ctx);
    if (!problems || count < 0) return fail(LAFF_E_ARG, "laff_fc_act_bn_strip_grouped: bad problem list");
    DeviceGuard g(ctx->device);
    auto kind = [](int act) { return act == LAFF_ACT_TANH || act == LAFF_ACT_SIGMOID ? 2 : (act == LAFF_ACT_RELU ? 1 : 0); };
    std::vector<char> done((size_t)count, 0);
    for (int i = 0; i < count; ++i) {
        const laff_fc_strip_problem& q = problems[i];
        if (q.N == 0) { done[i] = 1; continue; }
        if (!q.X || !q.img || !q.Y) return fail(LAFF_E_ARG, "laff_fc_act_bn_strip_grouped: problem %d has a null operand", i);
        if (q.N < 0 || q.D < 32 || (q.D & 31) || q.ldy < q.D || q.ldx < laff::FC_STRIP_K || (q.ldx & 3))
            return fail(LAFF_E_SHAPE, "laff_fc_act_bn_strip_grouped: problem %d: need D %% 32 == 0, ldx >= 512, ldx %% 4 == 0 (N=%d D=%d ldx=%d ldy=%d)",
                        i, q.N, q.D, q.ldx, q.ldy);
        if (q.act < LAFF_ACT_NONE || q.act > LAFF_ACT_SIGMOID) return fail(LAFF_E_ARG, "laff_fc_act_bn_strip_grouped: bad act %d", q.act);
        if (!aligned16(q.X) || !aligned16(q.img)) return fail(LAFF_E_ALIGN, "laff_fc_act_bn_strip_grouped: problem %d: 16-byte alignment", i);
        if ((long long)laff::FC_STRIP_ROWS * q.ldy * 4 >= (1ll << 31) || (long long)q.D * 2048 >= (1ll << 32))
            return fail(LAFF_E_UNSUPPORTED, "laff_fc_act_bn_strip_grouped: problem %d: output strip / image exceeds the 32-bit buffer range", i);
    }
    // one launch per (D, activation kind), up to MAX_GROUP problems each
    for (int i = 0; i < count; ++i) {
        if (done[i]) continue;
        laff::FcStripArgs fa{};
        fa.nblk = problems[i].D / 32;
        const int k = kind(problems[i].act);
        for (int j = i; j < count && fa.count < laff::MAX_GROUP; ++j) {
            const laff_fc_strip_problem& q = problems[j];
            if (done[j] || q.D != problems[i].D || kind(q.act) != k) continue;
            laff::FcStripProblem& fp = fa.p[fa.count++];
            fp.X = q.X; fp.img = q.img; fp.vec = (const float*)((const char*)q.img + laff::fc_strip_vec_offset(q.D));
            fp.Y = q.Y; fp.ldx = q.ldx; fp.ldy = q.ldy; fp.N = q.N;
            done[j] = 1;
        }
        HIP_TRY(laff::launch_fc_strip(fa, problems[i].act, ctx->stream));
    }
    return LAFF_OK;
}

int laff_fuse_packed(laff_ctx* ctx, const laff_plane* planes, int L, int N, int H, int d, const float* w, const float* b,
                     const float* gw, unsigned flags, float* E, float* attn_w, void* E16, int precision, float prescale);
int laff_fuse_packed_rank(laff_ctx* ctx, const laff_plane* planes, int L, int N, int H, int d, const float* w, const float* b,
                          const float* gw, unsigned flags, float* E, float* attn_w, void* E16, int precision, float prescale,
                          const laff_rank_side* rs);

int laff_fuse(laff_ctx* ctx, const laff_plane* planes, int L, int N, int H, int d, const float* w, const float* b,
              const float* gw, unsigned flags, float* E, float* attn_w) {
    return laff_fuse_packed(ctx, planes, L, N, H, d, w, b, gw, flags, E, attn_w, nullptr, LAFF_PREC_FP16, 1.0f);
}

// plane descriptors -> FuseArgs (shared by laff_fuse* and laff_plane_row_norms)
static int parse_planes(const laff_plane* planes, int L, int N, int H, int d, unsigned flags, laff::FuseArgs& a, bool& any_gather) {
    if (!planes) return fail(LAFF_E_ARG, "laff_fuse: null planes");
    if (L < 1 || L > laff::MAX_L) return fail(LAFF_E_SHAPE, "laff_fuse: L=%d outside [1,%d]", L, laff::MAX_L);
    if (N < 0 || H < 1 || d < 4 || (d & 3)) return fail(LAFF_E_SHAPE, "laff_fuse: need N>=0, H>=1, d%%4==0 (N=%d H=%d d=%d)", N, H, d);
    const bool nosplit = flags & LAFF_ATT_NO_SPLIT_HEAD;
    any_gather = false;
    for (int l = 0; l < L; ++l) {
        const laff_plane& p = planes[l];
        if ((p.scale == nullptr) != (p.shift == nullptr)) return fail(LAFF_E_ARG, "laff_fuse: plane %d scale/shift must come together", l);
        if (p.act < LAFF_ACT_NONE || p.act > LAFF_ACT_SIGMOID) return fail(LAFF_E_ARG, "laff_fuse: plane %d bad act %d", l, p.act);
        if (p.scale && (!aligned16(p.scale) || !aligned16(p.shift))) return fail(LAFF_E_ALIGN, "laff_fuse: plane %d affine not 16-byte aligned", l);
        a.rownorm[l] = p.row_scale;
        if (!p.src && p.wt) {                          // gather plane
            if (!p.indptr || !p.indices) return fail(LAFF_E_ARG, "laff_fuse: gather plane %d needs indptr / indices", l);
            if (p.tile || nosplit || d > 512) return fail(LAFF_E_UNSUPPORTED, "laff_fuse: gather plane %d needs split heads of d <= 512, not tiled", l);
            if (p.row_scale) return fail(LAFF_E_UNSUPPORTED, "laff_fuse: gather plane %d cannot take a row_scale (project the feature with laff_fc_gather_act_bn first)", l);
            if (p.dk < 1 || p.ldwt < H * d || (p.ldwt & 3)) return fail(LAFF_E_SHAPE, "laff_fuse: gather plane %d: dk=%d ldwt=%d (need >= %d, multiple of 4)", l, p.dk, p.ldwt, H * d);
            if (!aligned16(p.wt) || (p.bias && !aligned16(p.bias))) return fail(LAFF_E_ALIGN, "laff_fuse: gather plane %d not 16-byte aligned", l);
            a.g_indptr[l] = p.indptr; a.g_indices[l] = p.indices; a.g_values[l] = p.values; a.g_wt[l] = p.wt; a.g_bias[l] = p.bias;
            a.g_ldwt[l] = p.ldwt; a.g_dk[l] = p.dk;
            a.scale[l] = p.scale; a.shift[l] = p.shift; a.act[l] = p.act;
            any_gather = true;
            continue;
        }
        if (!p.src) return fail(LAFF_E_ARG, "laff_fuse: plane %d has null src", l);
        if (p.tile && nosplit) return fail(LAFF_E_UNSUPPORTED, "laff_fuse: tiled plane with NO_SPLIT_HEAD");
        const int need = p.tile ? d : (nosplit ? d : H * d);
        if (p.ld < need || (p.ld & 3)) return fail(LAFF_E_SHAPE, "laff_fuse: plane %d ld=%d (need >= %d, multiple of 4)", l, p.ld, need);
        if (!aligned16(p.src)) return fail(LAFF_E_ALIGN, "laff_fuse: plane %d not 16-byte aligned", l);
        a.src[l] = p.src; a.ld[l] = p.ld; a.tile[l] = p.tile; a.scale[l] = p.scale; a.shift[l] = p.shift; a.act[l] = p.act;
    }
    a.L = L; a.N = N; a.H = H; a.d = d; a.head_stride = nosplit ? 0 : d;
    a.flags = flags;
    return LAFF_OK;
}

int laff_fuse_packed(laff_ctx* ctx, const laff_plane* planes, int L, int N, int H, int d, const float* w, const float* b,
                     const float* gw, unsigned flags, float* E, float* attn_w, void* E16, int precision, float prescale) {
    return laff_fuse_packed_rank(ctx, planes, L, N, H, d, w, b, gw, flags, E, attn_w, E16, precision, prescale, nullptr);
}

int laff_fuse_packed_rank(laff_ctx* ctx, const laff_plane* planes, int L, int N, int H, int d, const float* w, const float* b,
                          const float* gw, unsigned flags, float* E, float* attn_w, void* E16, int precision, float prescale,
                          const laff_rank_side* rs) {
    CHECK_CTX(ctx);
    if (N == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (rs) {
        if (rs->side != 1 && rs->side != 2) return fail(LAFF_E_ARG, "laff_fuse_packed_rank: side must be 1 (text) or 2 (video)");
        if (!E16 || d > 512 || (flags & LAFF_ATT_NO_SPLIT_HEAD))
            return fail(LAFF_E_UNSUPPORTED, "laff_fuse_packed_rank: needs the 16-bit operand (E16) and split heads of d <= 512 (H=%d d=%d): use laff_rank_prepare", H, d);
        if (H > 1 && (!rs->partials || !rs->tickets)) return fail(LAFF_E_ARG, "laff_fuse_packed_rank: several heads need the partials / tickets scratch");
        if (!rs->band) return fail(LAFF_E_ARG, "laff_fuse_packed_rank: null band");
        if (rs->side == 1) {
            if (!rs->gt_col || !rs->Ev || !rs->s_gt64 || !rs->band_v || !rs->count || !rs->pairs)
                return fail(LAFF_E_ARG, "laff_fuse_packed_rank: the text side needs gt_col, Ev, s_gt64, band_v, count and pairs");
            if (rs->Nv < 1 || (long)N * 16 < rs->Nv)
                return fail(LAFF_E_UNSUPPORTED, "laff_fuse_packed_rank: %d text rows cannot finish the block maxima of %d videos: use laff_rank_prepare", N, rs->Nv);
            if (!aligned16(rs->Ev)) return fail(LAFF_E_ALIGN, "laff_fuse_packed_rank: Ev must be 16-byte aligned");
        }
    }
    if (E16 && precision != LAFF_PREC_FP16 && precision != LAFF_PREC_BF16)
        return fail(LAFF_E_UNSUPPORTED, "laff_fuse_packed: E16 is a single-plane operand (FP16 or BF16), got precision %d", precision);
    if (E16 && !aligned16(E16)) return fail(LAFF_E_ALIGN, "laff_fuse_packed: E16 must be 16-byte aligned");
    if (E16 && (flags & LAFF_ATT_JUST_AVERAGE)) return fail(LAFF_E_UNSUPPORTED, "laff_fuse_packed: JUST_AVERAGE output is not unit-norm");
    if (!planes || !E) return fail(LAFF_E_ARG, "laff_fuse: null planes/E");
    const bool javg = flags & LAFF_ATT_JUST_AVERAGE;
    if (!javg && (!w || !b)) return fail(LAFF_E_ARG, "laff_fuse: null w/b");
    if ((flags & LAFF_ATT_WITH_AVE) && !gw) return fail(LAFF_E_ARG, "laff_fuse: WITH_AVE needs gw");
    if (!aligned16(E) || (w && !aligned16(w))) return fail(LAFF_E_ALIGN, "laff_fuse: E/w must be 16-byte aligned");
    laff::FuseArgs a{};
    bool any_gather = false;
    if (int rc = parse_planes(planes, L, N, H, d, flags, a, any_gather)) return rc;
    a.head_major = any_gather ? 1 : 0;
    a.w = w; a.b = b; a.gw = gw; a.E = E; a.attn_w = attn_w;
    a.E16 = E16; a.e16_bf16 = precision == LAFF_PREC_BF16; a.e16_scale = prescale;
    if (rs) {
        a.rp_side = rs->side; a.rp_gt = rs->gt_col; a.rp_col0 = rs->col0; a.rp_Nv = rs->Nv; a.rp_Ev = rs->Ev; a.rp_sgt = rs->s_gt64;
        a.rp_band = rs->band; a.rp_band_v = rs->band_v; a.rp_count = rs->count; a.rp_pairs = rs->pairs;
        a.rp_part = rs->partials; a.rp_ticket = rs->tickets;
        // the same constants as laff_rank_prepare (rank.hip: launch_rank_prepare) for a single-plane operand
        a.rp_unit = precision == LAFF_PREC_BF16 ? 3.90625e-3f : 4.8828125e-4f;
        a.rp_cacc = (float)((double)H * d * 1.1920929e-7 + 9.5367432e-7);
    }
    DeviceGuard g(ctx->device);
    if (rs && H > 1) HIP_TRY(hipMemsetAsync(rs->tickets, 0, (size_t)N * sizeof(unsigned), ctx->stream));
    HIP_TRY(laff::launch_fuse(a, ctx->stream));
    return LAFF_OK;
}

int laff_plane_row_norms(laff_ctx* ctx, const laff_plane* planes, int L, int N, int H, int d, unsigned flags, float* out) {
    CHECK_CTX(ctx);
    if (N == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!out) return fail(LAFF_E_ARG, "laff_plane_row_norms: null out");
    laff::FuseArgs a{};
    bool any_gather = false;
    if (int rc = parse_planes(planes, L, N, H, d, flags, a, any_gather)) return rc;
    if (any_gather) return fail(LAFF_E_UNSUPPORTED, "laff_plane_row_norms: gather planes are not supported (project the feature first)");
    for (int l = 0; l < L; ++l)
        if (a.rownorm[l]) return fail(LAFF_E_ARG, "laff_plane_row_norms: plane %d already carries a row_scale", l);
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_plane_row_norms(a, out, ctx->stream));
    return LAFF_OK;
}

static int frame_fuse_grouped(laff_ctx* ctx, int count, const float* const* frames, const int* lens, const float* mask, int ldm, int B, int Fmax,
                              int d, const float* const* w, const float* const* b, const float* const* gw, unsigned flags, float* const* V) {
    CHECK_CTX(ctx);
    if (B == 0 || count == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (count < 0 || !frames || !w || !b || !V) return fail(LAFF_E_ARG, "laff_frame_fuse: null frames/w/b/V");
    if (B < 0 || Fmax < 1 || d < 4 || (d & 3) || d > 1024)
        return fail(LAFF_E_SHAPE, "laff_frame_fuse: need B>=0, Fmax>=1, d%%4==0, d<=1024 (B=%d Fmax=%d d=%d)", B, Fmax, d);
    if (flags & ~(unsigned)(LAFF_ATT_WITH_AVE | LAFF_ATT_MUL)) return fail(LAFF_E_UNSUPPORTED, "laff_frame_fuse: flags 0x%x", flags);
    if ((long)B * (count < 8 ? count : 8) > 0x7fffffffL) return fail(LAFF_E_SHAPE, "laff_frame_fuse: grid too large");
    DeviceGuard g(ctx->device);
    for (int i0 = 0; i0 < count; i0 += 8) {
        laff::FrameGroup grp{};
        grp.count = count - i0 < 8 ? count - i0 : 8;
        for (int i = 0; i < grp.count; ++i) {
            const int k = i0 + i;
            if (!frames[k] || !w[k] || !b[k] || !V[k]) return fail(LAFF_E_ARG, "laff_frame_fuse: feature %d has a null pointer", k);
            if ((flags & LAFF_ATT_WITH_AVE) && (!gw || !gw[k])) return fail(LAFF_E_ARG, "laff_frame_fuse: WITH_AVE needs gw");
            if (!aligned16(frames[k]) || !aligned16(w[k]) || !aligned16(V[k])) return fail(LAFF_E_ALIGN, "laff_frame_fuse: 16-byte alignment");
            grp.f[i] = laff::FrameArgs{frames[k], lens, B, Fmax, d, w[k], b[k], gw ? gw[k] : nullptr, flags, V[k], mask, ldm};
        }
        HIP_TRY(laff::launch_frame_fuse(grp, ctx->stream));
    }
    return LAFF_OK;
}

int laff_frame_fuse_grouped(laff_ctx* ctx, int count, const float* const* frames, const int* lens, int B, int Fmax, int d,
                            const float* const* w, const float* const* b, const float* const* gw, unsigned flags, float* const* V) {
    return frame_fuse_grouped(ctx, count, frames, lens, nullptr, 0, B, Fmax, d, w, b, gw, flags, V);
}

int laff_frame_fuse_grouped_mask(laff_ctx* ctx, int count, const float* const* frames, const float* mask, int ldm, int B, int Fmax, int d,
                                 const float* const* w, const float* const* b, const float* const* gw, unsigned flags, float* const* V) {
    if (B > 0 && count > 0 && (!mask || ldm < Fmax)) return fail(LAFF_E_ARG, "laff_frame_fuse_grouped_mask: mask must be (B, ldm >= Fmax)");
    return frame_fuse_grouped(ctx, count, frames, nullptr, mask, ldm, B, Fmax, d, w, b, gw, flags, V);
}

int laff_frame_fuse(laff_ctx* ctx, const float* frames, const int* lens, int B, int Fmax, int d, const float* w,
                    const float* b, const float* gw, unsigned flags, float* V) {
    return laff_frame_fuse_grouped(ctx, 1, &frames, lens, B, Fmax, d, &w, &b, gw ? &gw : nullptr, flags, &V);
}

int laff_packed_bytes(int N, int K, int precision, size_t* out) {
    if (!out || N < 0 || K < 0) return fail(LAFF_E_ARG, "laff_packed_bytes: bad args");
    if (precision < LAFF_PREC_FP32 || precision > LAFF_PREC_BF16X3) return fail(LAFF_E_ARG, "laff_packed_bytes: bad precision %d", precision);
    *out = (size_t)N * K * elem_size(precision) * (is_x3(precision) ? 2 : 1);
    return LAFF_OK;
}

int laff_pack_rows(laff_ctx* ctx, const float* E, int N, int H, int d, int lde, int normalize, float eps, float prescale,
                   int precision, void* out) {
    CHECK_CTX(ctx);
    if (N == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!E || !out) return fail(LAFF_E_ARG, "laff_pack_rows: null E/out");
    if (N < 0 || H < 1 || d < 1 || lde < H * d)
        return fail(LAFF_E_SHAPE, "laff_pack_rows: bad shape N=%d H=%d d=%d lde=%d", N, H, d, lde);
    if (precision < LAFF_PREC_FP32 || precision > LAFF_PREC_BF16X3) return fail(LAFF_E_ARG, "laff_pack_rows: bad precision %d", precision);
    const bool vec = !(d & 3) && !(lde & 3) && aligned16(E) && aligned16(out);
    if (!vec && precision != LAFF_PREC_FP32)
        return fail(LAFF_E_ALIGN, "laff_pack_rows: 16-bit output needs d%%4==0, lde%%4==0 and 16-byte aligned buffers (d=%d lde=%d)", d, lde);
    if (N == 0) return LAFF_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_pack_rows(E, N, H, d, lde, normalize, eps, prescale, precision, out, ctx->stream));
    return LAFF_OK;
}

static int sim_gemm_impl(laff_ctx* ctx, const char* who, const void* T, const void* V, int Nt, int Nv, int K, float scale,
                         int precision, float* S, int lds, const int* gt_col, int col0, const float* s_gt, int* count,
                         const double* s_gt64, const float* band_t, const float* band_v, unsigned* pairs, unsigned pair_cap) {
    CHECK_CTX(ctx);
    if (Nt == 0 || Nv == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!T || !V) return fail(LAFF_E_ARG, "%s: null T/V", who);
    if (precision < LAFF_PREC_FP32 || precision > LAFF_PREC_BF16X3) return fail(LAFF_E_ARG, "%s: bad precision %d", who, precision);
    const int esz = precision == LAFF_PREC_FP32 ? 4 : 2;
    if (Nt < 0 || Nv < 0 || K < 1 || ((long)K * esz) % 4)
        return fail(LAFF_E_SHAPE, "%s: K must be positive (and even for 16-bit operands) (Nt=%d Nv=%d K=%d)", who, Nt, Nv, K);
    if (!S && !gt_col) return fail(LAFF_E_ARG, "%s: nothing to produce (S and gt_col both null)", who);
    if (S && lds < Nv) return fail(LAFF_E_SHAPE, "%s: lds=%d < Nv=%d", who, lds, Nv);
    if (gt_col && !count) return fail(LAFF_E_ARG, "%s: gt_col needs count", who);
    if (gt_col && !s_gt && !s_gt64) return fail(LAFF_E_ARG, "%s: gt_col needs the ground-truth scores", who);
    pair_cap &= ~3u;            /* the resolve kernel reads the list four slots at a time: a ragged tail is never used */
    if (s_gt64 && (!gt_col || !band_t || !band_v || !pairs || pair_cap < 4))
        return fail(LAFF_E_ARG, "%s: the banded count needs gt_col, band_t, band_v and a pair list of >= 4 slots", who);
    if (!aligned16(T) || !aligned16(V)) return fail(LAFF_E_ALIGN, "%s: operands must be 16-byte aligned", who);
    if (s_gt64 && (!aligned16(gt_col) || !aligned16(s_gt64) || !aligned16(band_t) || !aligned16(band_v)))
        return fail(LAFF_E_ALIGN, "%s: gt_col, s_gt64, band_t and band_v must be 16-byte aligned (fetched in 16-byte groups)", who);
    laff::GemmArgs a{};
    a.R = T; a.C = V; a.nR = Nt; a.nC = Nv; a.K = K; a.ldR = K; a.ldC = K;
    const long planeT = (long)Nt * K * 2, planeV = (long)Nv * K * 2;
    if (is_x3(precision)) {
        // virtual K concatenation: lo*hi, hi*lo first (small terms), hi*hi last
        a.nseg = 3;
        a.segR[0] = planeT; a.segC[0] = 0;
        a.segR[1] = 0;      a.segC[1] = planeV;
        a.segR[2] = 0;      a.segC[2] = 0;
    } else {
        a.nseg = 1; a.segR[0] = a.segC[0] = 0;
    }
    a.out = S; a.ldo = lds; a.scale = scale;
    a.gt_col = gt_col; a.col0 = col0; a.s_gt = s_gt; a.count = gt_col ? count : nullptr;
    a.s_gt64 = s_gt64; a.band_r = band_t; a.band_c = band_v; a.pairs = pairs; a.pair_cap = pair_cap;
#ifdef LAFF_GEMM_TRACE
    if (const char* e = getenv("LAFF_GEMM_TRACE_PTR")) a.trace = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
    int mode = laff::GEMM_F32;
    if (precision == LAFF_PREC_FP16 || precision == LAFF_PREC_FP16X3) mode = laff::GEMM_F16;
    if (precision == LAFF_PREC_BF16 || precision == LAFF_PREC_BF16X3) mode = laff::GEMM_BF16;
    DeviceGuard g(ctx->device);
    // rows of K elements: 16-byte aligned rows take the direct-to-LDS paths (K bytes a multiple of 128: the fast one),
    // anything else is staged through registers with element-wise K bounds
    HIP_TRY(laff::launch_gemm_nt(a, mode, ((long)K * esz) % 16 == 0, ctx->stream));
    return LAFF_OK;
}

int laff_sim_gemm(laff_ctx* ctx, const void* T, const void* V, int Nt, int Nv, int K, float scale, int precision,
                  float* S, int lds, const int* gt_col, int col0, const float* s_gt, int* count) {
    if (gt_col && !s_gt) return fail(LAFF_E_ARG, "laff_sim_gemm: gt_col needs s_gt and count");
    return sim_gemm_impl(ctx, "laff_sim_gemm", T, V, Nt, Nv, K, scale, precision, S, lds, gt_col, col0, s_gt, count, nullptr, nullptr,
                         nullptr, nullptr, 0);
}

int laff_sim_gemm_banded(laff_ctx* ctx, const void* T, const void* V, int Nt, int Nv, int K, float scale, int precision, float* S,
                         int lds, const int* gt_col, int col0, const double* s_gt64, const float* band_t, const float* band_v,
                         int* count, unsigned* pairs, unsigned pair_cap) {
    if (!gt_col || !s_gt64) return fail(LAFF_E_ARG, "laff_sim_gemm_banded: null gt_col / s_gt64");
    return sim_gemm_impl(ctx, "laff_sim_gemm_banded", T, V, Nt, Nv, K, scale, precision, S, lds, gt_col, col0, nullptr, count, s_gt64,
                         band_t, band_v, pairs, pair_cap);
}

int laff_rank_prepare(laff_ctx* ctx, const float* Et, const float* Ev, const void* T, const void* V, int Nt, int Nv, int H, int d,
                      int precision, float prescale, const int* gt_col, int col0, double* s_gt64, float* band_t, float* band_v,
                      int* zero_count, unsigned* pairs) {
    CHECK_CTX(ctx);
    if (Nt == 0 && Nv == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if ((Nt > 0 && (!Et || !T || !gt_col || !s_gt64 || !band_t)) || (Nv > 0 && (!Ev || !V || !band_v)))
        return fail(LAFF_E_ARG, "laff_rank_prepare: null argument");
    if (precision < LAFF_PREC_FP32 || precision > LAFF_PREC_BF16X3) return fail(LAFF_E_ARG, "laff_rank_prepare: bad precision %d", precision);
    if (Nt < 0 || Nv < 0 || H < 1 || d < 4 || (d & 3)) return fail(LAFF_E_SHAPE, "laff_rank_prepare: need H >= 1, d %% 4 == 0 (Nt=%d Nv=%d H=%d d=%d)", Nt, Nv, H, d);
    if (!(prescale > 0.0f)) return fail(LAFF_E_ARG, "laff_rank_prepare: prescale must be positive");
    if ((Et && !aligned16(Et)) || (Ev && !aligned16(Ev)) || (T && !aligned16(T)) || (V && !aligned16(V)))
        return fail(LAFF_E_ALIGN, "laff_rank_prepare: embeddings and operands must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_rank_prepare(Et, Ev, T, V, Nt, Nv, H, d, precision, prescale, gt_col, col0, s_gt64, band_t, band_v, zero_count,
                                      pairs, 3, ctx->stream));
    return LAFF_OK;
}

int laff_rank_prepare_part(laff_ctx* ctx, int sides, const float* Et, const float* Ev, const void* T, const void* V, int Nt, int Nv, int H,
                           int d, int precision, float prescale, const int* gt_col, int col0, double* s_gt64, float* band_t, float* band_v,
                           int* zero_count, unsigned* pairs) {
    CHECK_CTX(ctx);
    if (sides != 1 && sides != 2) return fail(LAFF_E_ARG, "laff_rank_prepare_part: sides must be 1 (text rows) or 2 (video rows)");
    if ((sides == 1 && Nt == 0) || (sides == 2 && Nv == 0)) return LAFF_OK;
    if (sides == 1 && (!Et || !T || !gt_col || !s_gt64 || !band_t || (Nv > 0 && !Ev))) return fail(LAFF_E_ARG, "laff_rank_prepare_part: null argument (text side)");
    if (sides == 2 && (!Ev || !V || !band_v)) return fail(LAFF_E_ARG, "laff_rank_prepare_part: null argument (video side)");
    if (precision < LAFF_PREC_FP32 || precision > LAFF_PREC_BF16X3) return fail(LAFF_E_ARG, "laff_rank_prepare_part: bad precision %d", precision);
    if (Nt < 0 || Nv < 0 || H < 1 || d < 4 || (d & 3)) return fail(LAFF_E_SHAPE, "laff_rank_prepare_part: need H >= 1, d %% 4 == 0 (Nt=%d Nv=%d H=%d d=%d)", Nt, Nv, H, d);
    if (!(prescale > 0.0f)) return fail(LAFF_E_ARG, "laff_rank_prepare_part: prescale must be positive");
    if ((Et && !aligned16(Et)) || (Ev && !aligned16(Ev)) || (T && !aligned16(T)) || (V && !aligned16(V)))
        return fail(LAFF_E_ALIGN, "laff_rank_prepare_part: embeddings and operands must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_rank_prepare(Et, Ev, T, V, Nt, Nv, H, d, precision, prescale, gt_col, col0, s_gt64, band_t, band_v, zero_count,
                                      pairs, sides, ctx->stream));
    return LAFF_OK;
}

int laff_rank_prepare_emit(laff_ctx* ctx, int emit, const float* Et, const float* Ev, void* T, void* V, int Nt, int Nv, int H, int d,
                           int precision, float prescale, const int* gt_col, int col0, double* s_gt64, float* band_t, float* band_v,
                           int* zero_count, unsigned* pairs) {
    CHECK_CTX(ctx);
    if (emit < 1 || emit > 3) return fail(LAFF_E_ARG, "laff_rank_prepare_emit: emit must be 1 (T), 2 (V) or 3 (both)");
    if (precision != LAFF_PREC_FP16 && precision != LAFF_PREC_BF16)
        return fail(LAFF_E_UNSUPPORTED, "laff_rank_prepare_emit: single-plane 16-bit operands only (precision %d): call laff_pack_rows + laff_rank_prepare", precision);
    if (Nt == 0 || Nv == 0) return LAFF_OK;
    if (!Et || !Ev || !T || !V || !gt_col || !s_gt64 || !band_t || !band_v) return fail(LAFF_E_ARG, "laff_rank_prepare_emit: null argument");
    if (Nt < 0 || Nv < 0 || H < 1 || d < 4 || (d & 3)) return fail(LAFF_E_SHAPE, "laff_rank_prepare_emit: need H >= 1, d %% 4 == 0 (Nt=%d Nv=%d H=%d d=%d)", Nt, Nv, H, d);
    if (!(prescale > 0.0f)) return fail(LAFF_E_ARG, "laff_rank_prepare_emit: prescale must be positive");
    if (!aligned16(Et) || !aligned16(Ev) || !aligned16(T) || !aligned16(V))
        return fail(LAFF_E_ALIGN, "laff_rank_prepare_emit: embeddings and operands must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_rank_prepare(Et, Ev, T, V, Nt, Nv, H, d, precision, prescale, gt_col, col0, s_gt64, band_t, band_v, zero_count,
                                      pairs, 3, ctx->stream, emit));
    return LAFF_OK;
}

int laff_rank_export_pairs(laff_ctx* ctx, const double* s_gt64, int* count, float* S, int lds, int Nv, unsigned* pairs, unsigned pair_cap,
                           const int* bounds, int world, int col0, unsigned* out, unsigned cap, unsigned* fill) {
    CHECK_CTX(ctx);
    if (!s_gt64 || !count || !pairs || !bounds || !out || !fill) return fail(LAFF_E_ARG, "laff_rank_export_pairs: null argument");
    pair_cap &= ~3u;
    if (world < 1 || world > 16 || cap < 4 || (cap & 3) || pair_cap < 4) return fail(LAFF_E_SHAPE, "laff_rank_export_pairs: need 1 <= world <= 16, cap %% 4 == 0 (world=%d cap=%u)", world, cap);
    if (S && lds < Nv) return fail(LAFF_E_SHAPE, "laff_rank_export_pairs: lds=%d < Nv=%d", lds, Nv);
    if (!aligned16(out)) return fail(LAFF_E_ALIGN, "laff_rank_export_pairs: out must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_rank_export(s_gt64, count, S, lds, pairs, pair_cap, bounds, world, col0, out, cap, fill, ctx->stream));
    return LAFF_OK;
}

int laff_rank_resolve(laff_ctx* ctx, const float* Et, const float* Ev, int Nt, int Nv, int H, int d, const double* s_gt64,
                      int* count, float* S, int lds, unsigned* pairs, unsigned pair_cap) {
    CHECK_CTX(ctx);
    if (Nt == 0 || Nv == 0) return LAFF_OK;                 /* empty problem: nothing was listed */
    if (!Et || !Ev || !s_gt64 || !count || !pairs) return fail(LAFF_E_ARG, "laff_rank_resolve: null argument");
    pair_cap &= ~3u;            /* as laff_sim_gemm_banded: whole groups of four slots */
    if (Nt < 0 || Nv < 0 || H < 1 || d < 4 || (d & 3) || pair_cap < 4) return fail(LAFF_E_SHAPE, "laff_rank_resolve: bad shape");
    if (S && lds < Nv) return fail(LAFF_E_SHAPE, "laff_rank_resolve: lds=%d < Nv=%d", lds, Nv);
    if (!aligned16(Et) || !aligned16(Ev)) return fail(LAFF_E_ALIGN, "laff_rank_resolve: embeddings must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_rank_resolve(Et, Ev, Nt, Nv, H, d, s_gt64, count, S, lds, pairs, pair_cap, ctx->stream));
    return LAFF_OK;
}

int laff_rank_resolve_metrics(laff_ctx* ctx, const float* Et, const float* Ev, int Nt, int Nv, int H, int d, const double* s_gt64,
                              int* count, float* S, int lds, unsigned* pairs, unsigned pair_cap, int base, int* ranks_out, double* out8,
                              int synchronous) {
    CHECK_CTX(ctx);
    if (!out8) return fail(LAFF_E_ARG, "laff_rank_resolve_metrics: null out8");
    if (Nt < 1 || Nv < 1) return fail(LAFF_E_SHAPE, "laff_rank_resolve_metrics: Nt=%d Nv=%d (the metrics of an empty query set are undefined)", Nt, Nv);
    if (!Et || !Ev || !s_gt64 || !count || !pairs) return fail(LAFF_E_ARG, "laff_rank_resolve_metrics: null argument");
    pair_cap &= ~3u;
    if (H < 1 || d < 4 || (d & 3) || pair_cap < 4) return fail(LAFF_E_SHAPE, "laff_rank_resolve_metrics: bad shape");
    if (S && lds < Nv) return fail(LAFF_E_SHAPE, "laff_rank_resolve_metrics: lds=%d < Nv=%d", lds, Nv);
    if (!aligned16(Et) || !aligned16(Ev)) return fail(LAFF_E_ALIGN, "laff_rank_resolve_metrics: embeddings must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    if (int rc = metrics_buffers(ctx)) return rc;
    const size_t si = synchronous ? 0 : 1 + ctx->metrics_slot++ % (METRIC_SLOTS - 1);
    double* slot = ctx->d_metrics + 8 * si;
    unsigned* ticket = (unsigned*)(ctx->d_mscratch + si * laff::rank_metrics_scratch_bytes() + laff::rank_resolve_ticket_offset());
    // Pinned (hipHostMalloc'ed / registered) result buffers are addressable from the device: the finishing workgroup stores the 64 bytes
    // there itself and no copy node follows the launch.  Anything else gets the copy.
    double* host8 = nullptr;
    if (!synchronous) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, out8, 0) == hipSuccess && dp) host8 = (double*)dp;
        else (void)hipGetLastError();
    }
    HIP_TRY(laff::launch_rank_resolve(Et, Ev, Nt, Nv, H, d, s_gt64, count, S, lds, pairs, pair_cap, ctx->stream, Nt, base, ranks_out, slot,
                                      host8, ticket));
    if (synchronous) {
        HIP_TRY(hipMemcpyAsync(ctx->h_metrics, slot, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->h_metrics[7] != 0.0)
            return fail(LAFF_E_ARG, "laff_rank_resolve_metrics: a rank < 1 was found (the pair list of laff_sim_gemm_banded overflowed, or the "
                                    "counts are corrupt)");
        for (int i = 0; i < 7; ++i) out8[i] = ctx->h_metrics[i];
        out8[7] = 0.0;
    } else if (!host8) {
        HIP_TRY(hipMemcpyAsync(out8, slot, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    return LAFF_OK;
}

int laff_gather_gt(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, const int* gt_col, int col0, float* s_gt) {
    CHECK_CTX(ctx);
    if (Nt == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!S || !gt_col || !s_gt) return fail(LAFF_E_ARG, "laff_gather_gt: null argument");
    if (Nt < 0 || Nv < 0 || lds < Nv) return fail(LAFF_E_SHAPE, "laff_gather_gt: bad shape");
    if (Nt == 0) return LAFF_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_gather_gt(S, Nt, Nv, lds, gt_col, col0, s_gt, ctx->stream));
    return LAFF_OK;
}

int laff_rank_count(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, const int* gt_col, int col0,
                    const float* s_gt, int* count, int accumulate) {
    CHECK_CTX(ctx);
    if (Nt == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!S || !gt_col || !s_gt || !count) return fail(LAFF_E_ARG, "laff_rank_count: null argument");
    if (Nt < 0 || Nv < 0 || lds < Nv) return fail(LAFF_E_SHAPE, "laff_rank_count: bad shape");
    if (Nt == 0) return LAFF_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_rank_count(S, Nt, Nv, lds, gt_col, col0, s_gt, count, accumulate, ctx->stream));
    return LAFF_OK;
}

int laff_topk_rows(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, int K, int* idx_out, float* val_out) {
    CHECK_CTX(ctx);
    if (Nt == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!S || !idx_out || !val_out) return fail(LAFF_E_ARG, "laff_topk_rows: null argument");
    if (Nt < 0 || Nv < 1 || lds < Nv || K < 1 || K > Nv || K > 8192) return fail(LAFF_E_SHAPE, "laff_topk_rows: need 1 <= K <= min(Nv, 8192) (Nt=%d Nv=%d K=%d)", Nt, Nv, K);
    {
        const size_t kp = K <= 64 ? 64 : (K <= 512 ? 512 : (K <= 2048 ? 2048 : (K <= 4096 ? 4096 : 8192)));
        if ((size_t)((Nv + 3) & ~3) * 4 + kp * 8 + 256 * 4 + 16 > 160 * 1024)
            return fail(LAFF_E_UNSUPPORTED, "laff_topk_rows: Nv=%d with K=%d does not fit the LDS-resident row (%zu columns at most: split the "
                        "columns and merge the per-block lists, as laff_amd.ops.topk_rows does)", Nv, K, (160 * 1024 - kp * 8 - 1040) / 4);
    }
    if (Nt == 0) return LAFF_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_topk_rows(S, Nt, Nv, lds, K, idx_out, val_out, ctx->stream));
    return LAFF_OK;
}

int laff_v2t_count(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, const int* grp_off, const int* grp_idx,
                   int max_group, int* count) {
    CHECK_CTX(ctx);
    if (Nt == 0 || Nv == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!S || !grp_off || !grp_idx || !count) return fail(LAFF_E_ARG, "laff_v2t_count: null argument");
    if (Nt < 0 || Nv < 0 || lds < Nv || max_group < 0) return fail(LAFF_E_SHAPE, "laff_v2t_count: bad shape");
    if (Nt == 0 || Nv == 0 || max_group == 0) return LAFF_OK;
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_v2t_count(S, Nt, Nv, lds, grp_off, grp_idx, max_group, count, ctx->stream));
    return LAFF_OK;
}

int laff_v2t_count_exact(laff_ctx* ctx, const float* S, int Nt, int Nv, int lds, const int* grp_off, const int* grp_idx,
                         int max_group, const float* Et, const float* Ev, int H, int d, const double* s_gt64,
                         const float* band_t, const float* band_v, int* count, unsigned* list, unsigned list_cap) {
    CHECK_CTX(ctx);
    if (Nt == 0 || Nv == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!S || !grp_off || !grp_idx || !count || !Et || !Ev || !s_gt64 || !band_t || !band_v || !list)
        return fail(LAFF_E_ARG, "laff_v2t_count_exact: null argument");
    if (Nt < 0 || Nv < 0 || lds < Nv || max_group < 0 || H < 1 || d < 4 || (d & 3))
        return fail(LAFF_E_SHAPE, "laff_v2t_count_exact: bad shape (Nt=%d Nv=%d lds=%d H=%d d=%d: d must be a multiple of 4)", Nt, Nv, lds, H, d);
    if (!aligned16(Et) || !aligned16(Ev)) return fail(LAFF_E_ALIGN, "laff_v2t_count_exact: embeddings must be 16-byte aligned");
    if (list_cap < 1) return fail(LAFF_E_SHAPE, "laff_v2t_count_exact: list_cap must be >= 1");
    DeviceGuard g(ctx->device);
    if (max_group == 0) {
        HIP_TRY(hipMemsetAsync(count, 0, (size_t)Nt * sizeof(int), ctx->stream));
        HIP_TRY(hipMemsetAsync(list, 0, 16, ctx->stream));
        return LAFF_OK;
    }
    HIP_TRY(laff::launch_v2t_count_exact(S, Nt, Nv, lds, grp_off, grp_idx, max_group, Et, Ev, H, d, s_gt64, band_t, band_v, count, list,
                                         list_cap, ctx->stream));
    return LAFF_OK;
}

int laff_row_dot_gt(laff_ctx* ctx, const void* T, const void* V, int Nt, int Nv, int K, float scale, int precision,
                    const int* gt_col, int col0, float* s_gt, int* zero_count) {
    CHECK_CTX(ctx);
    if (Nt == 0) return LAFF_OK;                 /* empty problem: nothing to launch, pointers may be null */
    if (!T || !V || !gt_col || !s_gt) return fail(LAFF_E_ARG, "laff_row_dot_gt: null argument");
    if (precision < LAFF_PREC_FP16 || precision > LAFF_PREC_BF16X3) return fail(LAFF_E_UNSUPPORTED, "laff_row_dot_gt: 16-bit precisions only (got %d)", precision);
    if (Nt < 0 || Nv < 0 || K < 2 || (K & 1)) return fail(LAFF_E_SHAPE, "laff_row_dot_gt: K must be positive and even (K=%d)", K);
    if (!aligned16(T) || !aligned16(V)) return fail(LAFF_E_ALIGN, "laff_row_dot_gt: operands must be 16-byte aligned");
    if (Nt == 0) return LAFF_OK;
    const int bf16 = (precision == LAFF_PREC_BF16 || precision == LAFF_PREC_BF16X3);
    DeviceGuard g(ctx->device);
    HIP_TRY(laff::launch_row_dot_gt(T, V, Nt, Nv, K, bf16, is_x3(precision) ? 1 : 0, scale, gt_col, col0, s_gt, zero_count, ctx->stream));
    return LAFF_OK;
}

int laff_rank_metrics_async(laff_ctx* ctx, const int* rank1, int Nq, int base, int* ranks_out, double* out8) {
    CHECK_CTX(ctx);
    if (!rank1 || !out8) return fail(LAFF_E_ARG, "laff_rank_metrics_async: null argument");
    if (Nq < 1) return fail(LAFF_E_SHAPE, "laff_rank_metrics_async: Nq=%d", Nq);
    DeviceGuard g(ctx->device);
    if (int rc = metrics_buffers(ctx)) return rc;
    const size_t si = 1 + ctx->metrics_slot++ % (METRIC_SLOTS - 1);                                  // (slot 0: the synchronous call)
    double* slot = ctx->d_metrics + 8 * si;
    // a pinned result buffer is addressable from the device: the finishing workgroup stores the 64 bytes there itself, no copy node
    double* host8 = nullptr;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, out8, 0) == hipSuccess && dp) host8 = (double*)dp;
    else (void)hipGetLastError();
    HIP_TRY(laff::launch_rank_metrics(rank1, Nq, base, ranks_out, slot, slot + 7,
                                      (unsigned*)(ctx->d_mscratch + si * laff::rank_metrics_scratch_bytes()), ctx->stream, host8));
    if (!host8) HIP_TRY(hipMemcpyAsync(out8, slot, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return LAFF_OK;
}

int laff_rank_metrics(laff_ctx* ctx, const int* rank1, int Nq, int base, int* ranks_out, double out7[7]) {
    CHECK_CTX(ctx);
    if (!rank1 || !out7) return fail(LAFF_E_ARG, "laff_rank_metrics: null argument");
    if (Nq < 1) return fail(LAFF_E_SHAPE, "laff_rank_metrics: Nq=%d", Nq);
    DeviceGuard g(ctx->device);
    if (int rc = metrics_buffers(ctx)) return rc;
    HIP_TRY(laff::launch_rank_metrics(rank1, Nq, base, ranks_out, ctx->d_metrics, ctx->d_metrics + 7, (unsigned*)ctx->d_mscratch, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_metrics, ctx->d_metrics, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->h_metrics[7] != 0.0)
        return fail(LAFF_E_ARG, "laff_rank_metrics: a rank < 1 was found (ranks must be 1-based; a poisoned count also means the pair "
                                "list of laff_sim_gemm_banded overflowed)");
    for (int i = 0; i < 7; ++i) out7[i] = ctx->h_metrics[i];
    return LAFF_OK;
}

}  // extern "C"
