/* A host that is not Python: the exact-rank tail of the hot path through the C ABI of liblaff_hip.so alone (include/laff_hip.h), the way
 * a maintainer would bind it from C.  No torch, no Python: hipMalloc'ed buffers and plain pointers.
 *
 *   laff_host <problem.bin> <out.bin>
 * problem.bin: int32 {Nt, Nv, H, d} | float32 Et[Nt*H*d] | float32 Ev[Nv*H*d] | int32 gt[Nt]     (unit-norm embeddings, as the towers
 *              leave them -- what /root/reference/model/model.py:1003-1016 scores and predictor.py:232-246 ranks)
 * out.bin:     int32 ranks[Nt] | float64 metrics[7] | float32 S[Nt*Nv]
 * Steps: laff_rank_prepare_emit (both fp16 operands produced on the way) -> laff_sim_gemm_banded -> laff_rank_resolve_metrics.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "laff_hip.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define LAFF(x) do { int rc_ = (x); if (rc_ != LAFF_OK) { fprintf(stderr, "%s: %s (rc=%d)\n", #x, laff_last_error(), rc_); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s problem.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int hdr[4];
    if (fread(hdr, sizeof(int), 4, f) != 4) return 1;
    const int Nt = hdr[0], Nv = hdr[1], H = hdr[2], d = hdr[3], K = H * d;
    float* hEt = (float*)malloc((size_t)Nt * K * 4);
    float* hEv = (float*)malloc((size_t)Nv * K * 4);
    int* hgt = (int*)malloc((size_t)Nt * 4);
    if (fread(hEt, 4, (size_t)Nt * K, f) != (size_t)Nt * K || fread(hEv, 4, (size_t)Nv * K, f) != (size_t)Nv * K ||
        fread(hgt, 4, (size_t)Nt, f) != (size_t)Nt) { fprintf(stderr, "short problem file\n"); return 1; }
    fclose(f);
    if (laff_abi_version() != LAFF_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", laff_abi_version(), LAFF_ABI_VERSION); return 1; }

    hipStream_t st;
    HIP(hipStreamCreate(&st));
    laff_ctx* ctx;
    LAFF(laff_ctx_create(0, st, &ctx));

    const unsigned cap = 1u << 20;                                  /* slots of the in-band pair list */
    const size_t nbv = (size_t)((Nv + 3) & ~3) + (size_t)(Nv + 63) / 64 + 4;
    float *Et, *Ev, *S, *band_t, *band_v;
    void *T, *V;
    int *gt, *count, *ranks;
    double* s_gt64;
    unsigned* pairs;
    HIP(hipMalloc((void**)&Et, (size_t)Nt * K * 4));
    HIP(hipMalloc((void**)&Ev, (size_t)Nv * K * 4));
    HIP(hipMalloc(&T, (size_t)Nt * K * 2 + 16));
    HIP(hipMalloc(&V, (size_t)Nv * K * 2 + 16));
    HIP(hipMalloc((void**)&S, (size_t)Nt * Nv * 4));
    HIP(hipMalloc((void**)&gt, (size_t)Nt * 4 + 16));
    HIP(hipMalloc((void**)&count, (size_t)Nt * 4));
    HIP(hipMalloc((void**)&ranks, (size_t)Nt * 4));
    HIP(hipMalloc((void**)&s_gt64, (size_t)(Nt + 2) * 8));
    HIP(hipMalloc((void**)&band_t, (size_t)(Nt + 4) * 4));
    HIP(hipMalloc((void**)&band_v, nbv * 4));
    HIP(hipMalloc((void**)&pairs, (size_t)(4 + 2 * (size_t)cap) * 4));
    HIP(hipMemcpyAsync(Et, hEt, (size_t)Nt * K * 4, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(Ev, hEv, (size_t)Nv * K * 4, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(gt, hgt, (size_t)Nt * 4, hipMemcpyHostToDevice, st));

    const float prescale = 1.0f;
    /* exact ground-truth scores, error bands, cleared accumulators -- and both fp16 operands (loss.py:8-13 leaves unit-norm rows) */
    LAFF(laff_rank_prepare_emit(ctx, 3, Et, Ev, T, V, Nt, Nv, H, d, LAFF_PREC_FP16, prescale, gt, 0, s_gt64, band_t, band_v, count, pairs));
    /* S = T V^T / H with the banded count in the epilogue (model.py:1003-1016 + the count form of predictor.py:232-244) */
    LAFF(laff_sim_gemm_banded(ctx, T, V, Nt, Nv, K, 1.0f / ((float)H * prescale * prescale), LAFF_PREC_FP16, S, Nv, gt, 0, s_gt64, band_t, band_v,
                              count, pairs, cap));
    /* exact re-score of the pairs inside the band, ranks = count + 1, evaluation.eval's seven numbers (evaluation.py:92-109) */
    double out8[8];
    LAFF(laff_rank_resolve_metrics(ctx, Et, Ev, Nt, Nv, H, d, s_gt64, count, S, Nv, pairs, cap, 1, ranks, out8, /*synchronous*/1));

    int* hr = (int*)malloc((size_t)Nt * 4);
    float* hS = (float*)malloc((size_t)Nt * Nv * 4);
    HIP(hipMemcpy(hr, ranks, (size_t)Nt * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(hS, S, (size_t)Nt * Nv * 4, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fwrite(hr, 4, (size_t)Nt, f);
    fwrite(out8, 8, 7, f);
    fwrite(hS, 4, (size_t)Nt * Nv, f);
    fclose(f);
    printf("R@1 %.4f R@5 %.4f R@10 %.4f MedR %.1f meanr %.4f mir %.6f\n", out8[0], out8[1], out8[2], out8[3], out8[4], out8[5]);
    LAFF(laff_ctx_destroy(ctx));
    return 0;
}
