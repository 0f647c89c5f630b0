// rank.hip -- ranking kernels (gfx950): the argsort + label-matrix loop of /root/reference/predictor.py:232-244
// (text->video) and :262-270 (video->text) restated as counts, so no O(Nt*Nv) index or label matrix exists:
//   position(t, g) = 1 + #{ c != g : S[t,c] > S[t,g] }.
// HBM-bound: S is read once, 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

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
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
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
