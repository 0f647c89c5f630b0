// loss.hip -- per-head margin ranking loss on the in-batch cosine score matrix, forward + backward
// (/root/reference/loss.py:68-135 MarginRankingLoss, summed over heads as in
//  model/model.py:2032-2048 `for each in range(H): loss += criterion(txt[:, each], vis[:, each])`).
//
//   scores = l2norm(im) . l2norm(s)^T            (loss.py:30-34; rows = videos, columns = captions)
//   cost_s [i][j] = max(0, margin + scores[i][j] - scores[i][i])   ('i2t' / 'bidir'), diagonal cleared
//   cost_im[i][j] = max(0, margin + scores[i][j] - scores[j][j])   ('t2i' / 'bidir'), diagonal cleared
//   max_violation: cost_s -> max over j, cost_im -> max over i;  loss = sum (or mean) of both.
//
// The B x B x d contractions (scores, and the two gradient products dS . S^ and dS^T . I^) run on the fp32 MFMA GEMM of
// gemm_nt.hip (grouped over heads); this file holds the memory-bound pieces around them: row normalisation (+ the
// transposed copies the NT GEMM needs as its column operand), the hinge / max-violation reduction that also emits
// dLoss/dScores, and the backward of the row normalisation.  Batches are small (B ~ 128..1024): everything here is
// latency-bound and sized as one wave per row / one workgroup per head.
#include "kernels.h"
#include "wave_reduce.h"

namespace laff {

__device__ __forceinline__ float wave_sum(float v) { return wave_allsum(v); }      // (wave_reduce.h: the butterfly's bits, no LDS crossbar)

// grid (B, H, 2): z = 0 captions (s), z = 1 videos (im); one wave per row.
// XH[z][h][b][dp] = x / (|x| + eps + 1e-14),  XHT[z][h][k][Bp] its transpose,  nrm[z][h][b] = |x|,  npr = |x| + eps + 1e-14
__global__ __launch_bounds__(64) void loss_normalize_kernel(const float* __restrict__ s, const float* __restrict__ im, int B, int H,
                                                            int d, int dp, int Bp, float eps, float* __restrict__ XH,
                                                            float* __restrict__ XHT, float* __restrict__ nrm, float* __restrict__ npr) {
    const int b = blockIdx.x, h = blockIdx.y, z = blockIdx.z, lane = threadIdx.x;
    const float* x = (z ? im : s) + ((long)b * H + h) * d;
    float ss = 0.0f;
    for (int k = lane; k < d; k += 64) ss = fmaf(x[k], x[k], ss);
    const float r = sqrtf(wave_sum(ss));
    const float n = r + eps + 1e-14f;
    float* xh = XH + (((long)z * H + h) * B + b) * dp;
    float* xt = XHT + ((long)z * H + h) * d * Bp + b;
    for (int k = lane; k < dp; k += 64) {
        const float v = k < d ? x[k] / n : 0.0f;
        xh[k] = v;
        if (k < d) xt[(long)k * Bp] = v;
    }
    if (lane == 0) {
        nrm[((long)z * H + h) * B + b] = r;
        npr[((long)z * H + h) * B + b] = n;
    }
}

struct MarginArgs {
    const float* S;       // [H][B][Bp] scores (rows = videos, columns = captions)
    float* dS;            // [H][B][Bp] dLoss/dScores
    float* dST;           // [H][B][Bp] its transpose
    float* loss_h;        // [H]
    int B, Bp, H;
    float margin;
    int max_violation, use_s, use_im;
    float g_s, g_im;      // weight of one cost term in the total (1, or 1/count for cost_style = 'mean')
};

// one 1024-thread workgroup per head
__global__ __launch_bounds__(1024) void margin_reduce_kernel(MarginArgs a) {
    extern __shared__ float sh[];
    const int B = a.B, Bp = a.Bp, h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* S = a.S + (long)h * B * Bp;
    float* dS = a.dS + (long)h * B * Bp;
    float* dST = a.dST + (long)h * B * Bp;
    float* diag = sh;                 // [B]
    float* dacc = sh + B;             // [B]  gradient collected on the diagonal
    float* red = sh + 2 * B;          // [16]
    for (int i = tid; i < B; i += 1024) {
        diag[i] = S[(long)i * Bp + i];
        dacc[i] = 0.0f;
    }
    __syncthreads();
    float loss = 0.0f;
    if (!a.max_violation) {
        for (long e = tid; e < (long)B * B; e += 1024) {
            const int i = (int)(e / B), j = (int)(e % B);
            float g = 0.0f;
            if (i != j) {
                const float sc = S[(long)i * Bp + j];
                if (a.use_s) {
                    const float c = a.margin + sc - diag[i];
                    if (c > 0.0f) { loss += a.g_s * c; g += a.g_s; atomicAdd(&dacc[i], -a.g_s); }
                }
                if (a.use_im) {
                    const float c = a.margin + sc - diag[j];
                    if (c > 0.0f) { loss += a.g_im * c; g += a.g_im; atomicAdd(&dacc[j], -a.g_im); }
                }
            }
            dS[(long)i * Bp + j] = g;
        }
        __syncthreads();
    } else {
        for (long e = tid; e < (long)B * Bp; e += 1024) dS[e] = 0.0f;
        __syncthreads();
        if (a.use_s) {                               // hardest caption of every video: wave per row
            for (int i = wave; i < B; i += 16) {
                float best = 0.0f;
                int bj = -1;
                for (int j = lane; j < B; j += 64) {
                    const float c = (j == i) ? 0.0f : fmaxf(a.margin + S[(long)i * Bp + j] - diag[i], 0.0f);
                    if (c > best) { best = c; bj = j; }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {    // max, ties -> smallest column (torch.max returns the first maximum)
                    const float ob = __shfl_xor(best, o);
                    const int oj = __shfl_xor(bj, o);
                    if (ob > best || (ob == best && oj >= 0 && (bj < 0 || oj < bj))) { best = ob; bj = oj; }
                }
                if (lane == 0 && bj >= 0) {
                    loss += a.g_s * best;
                    atomicAdd(&dS[(long)i * Bp + bj], a.g_s);
                    atomicAdd(&dacc[i], -a.g_s);
                }
            }
        }
        if (a.use_im) {                              // hardest video of every caption: thread per column (coalesced rows)
            for (int j = tid; j < B; j += 1024) {
                float best = 0.0f;
                int bi = -1;
                for (int i = 0; i < B; ++i) {
                    const float c = (i == j) ? 0.0f : fmaxf(a.margin + S[(long)i * Bp + j] - diag[j], 0.0f);
                    if (c > best) { best = c; bi = i; }
                }
                if (bi >= 0) {
                    loss += a.g_im * best;
                    atomicAdd(&dS[(long)bi * Bp + j], a.g_im);
                    atomicAdd(&dacc[j], -a.g_im);
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < B; i += 1024) dS[(long)i * Bp + i] = dacc[i];
    // block sum of the loss
    loss = wave_sum(loss);
    if (lane == 0) red[wave] = loss;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += red[w];
        a.loss_h[h] = t;
    }
    __syncthreads();
    for (long e = tid; e < (long)B * B; e += 1024) {
        const int i = (int)(e / B), j = (int)(e % B);
        dST[(long)j * Bp + i] = dS[(long)i * Bp + j];
    }
}

__global__ void loss_sum_heads_kernel(const float* __restrict__ loss_h, int H, float* __restrict__ loss) {
    float t = 0.0f;
    for (int h = 0; h < H; ++h) t += loss_h[h];      // the reference adds the heads in order (model/model.py:2037-2039)
    loss[0] = t;
}

// backward of x^ = x / (|x| + eps'):  dx = g / n' - x^ (x^ . g) / |x|.  grid (B, H, 2), one wave per row.
__global__ __launch_bounds__(64) void loss_normalize_bwd_kernel(const float* __restrict__ XH, const float* __restrict__ G,
                                                                const float* __restrict__ nrm, const float* __restrict__ npr, int B,
                                                                int H, int d, int dp, float* __restrict__ d_s, float* __restrict__ d_im) {
    const int b = blockIdx.x, h = blockIdx.y, z = blockIdx.z, lane = threadIdx.x;
    float* out = z ? d_im : d_s;
    if (!out) return;
    const long row = (((long)z * H + h) * B + b);
    const float* xh = XH + row * dp;
    const float* g = G + row * dp;
    float dot = 0.0f;
    for (int k = lane; k < d; k += 64) dot = fmaf(xh[k], g[k], dot);
    dot = wave_sum(dot);
    const float n = npr[row], r = nrm[row];
    const float c = dot / r;
    float* o = out + ((long)b * H + h) * d;
    for (int k = lane; k < d; k += 64) o[k] = g[k] / n - xh[k] * c;
}

hipError_t launch_loss_normalize(const float* s, const float* im, int B, int H, int d, int dp, int Bp, float eps, float* XH,
                                 float* XHT, float* nrm, float* npr, hipStream_t st) {
    hipLaunchKernelGGL(loss_normalize_kernel, dim3(B, H, 2), dim3(64), 0, st, s, im, B, H, d, dp, Bp, eps, XH, XHT, nrm, npr);
    return hipGetLastError();
}

hipError_t launch_margin_reduce(const float* S, float* dS, float* dST, float* loss_h, float* loss, int B, int Bp, int H,
                                float margin, int max_violation, int use_s, int use_im, float g_s, float g_im, hipStream_t st) {
    MarginArgs a{S, dS, dST, loss_h, B, Bp, H, margin, max_violation, use_s, use_im, g_s, g_im};
    hipLaunchKernelGGL(margin_reduce_kernel, dim3(H), dim3(1024), (2 * B + 16) * sizeof(float), st, a);
    hipLaunchKernelGGL(loss_sum_heads_kernel, dim3(1), dim3(1), 0, st, loss_h, H, loss);
    return hipGetLastError();
}

hipError_t launch_loss_normalize_bwd(const float* XH, const float* G, const float* nrm, const float* npr, int B, int H, int d,
                                     int dp, float* d_s, float* d_im, hipStream_t st) {
    hipLaunchKernelGGL(loss_normalize_bwd_kernel, dim3(B, H, 2), dim3(64), 0, st, XH, G, nrm, npr, B, H, d, dp, d_s, d_im);
    return hipGetLastError();
}

}  // namespace laff
