// The eight XCDs of an MI355X run at clocks of their own (amd-smi metric -c shows 1,590 .. 1,735 MHz side by side under one load).  A
// launch that gives every CU the same share of work therefore ends when the slowest XCD does.  This probe measures what a share
// in proportion to the XCD's clock returns when the package sits at its power cap: one persistent workgroup per CU runs n[b] rounds of
// 16 dense MFMAs; per workgroup it records the XCD, wall-clock (100 MHz) start / end and shader cycles.
//     hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe/xcdprobe.hip -o xcdprobe && ./xcdprobe [rounds] [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Rec { unsigned long long xcc, t0, t1, cycles; };

__global__ __launch_bounds__(256, 1) void work(Rec* rec, const int* rounds, unsigned* sink) {
    extern __shared__ char smem[];
    f32x16 acc[4];
    // fp16 operands of magnitude ~2^-6 with lane-dependent mantissas (zeros would switch next to nothing)
    const unsigned h = 0x2400u | ((threadIdx.x * 37u) & 0x3ffu), g = 0xa400u | ((threadIdx.x * 91u) & 0x3ffu);
    u32x4 a = {h | (g << 16), g | (h << 16), h | (h << 16), g | (g << 16)}, b = {g | (h << 16), h | (g << 16), g | (g << 16), h | (h << 16)};
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    asm volatile("" : "+v"(a), "+v"(b));
    const int n = rounds[blockIdx.x];
    const unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    unsigned s = 0;
    for (int i = 0; i < 4; ++i) s += (unsigned)acc[i][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) rec[blockIdx.x] = Rec{(unsigned long long)(__builtin_amdgcn_s_getreg(0xf814) & 15u), t0, t1, c1 - c0};
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4000, L = argc > 2 ? atoi(argv[2]) : 600, G = 256;
    Rec* d; unsigned* sink; int* dn;
    hipMalloc(&d, G * sizeof(Rec)); hipMalloc(&sink, G * 256 * 4); hipMalloc(&dn, G * 4);
    hipFuncSetAttribute((const void*)work, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    std::vector<int> n(G, N);
    std::vector<double> w(8, 1.0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int phase = 0; phase < 4; ++phase) {
        hipMemcpy(dn, n.data(), G * 4, hipMemcpyHostToDevice);
        for (int i = 0; i < L / 2; ++i) hipLaunchKernelGGL(work, dim3(G), dim3(256), 100 * 1024, 0, d, dn, sink);
        hipEventRecord(e0);
        for (int i = 0; i < L / 2; ++i) hipLaunchKernelGGL(work, dim3(G), dim3(256), 100 * 1024, 0, d, dn, sink);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<Rec> r(G);
        hipMemcpy(r.data(), d, G * sizeof(Rec), hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (auto& x : r) { tmin = std::min(tmin, x.t0); tmax = std::max(tmax, x.t1); }
        double clk[8] = {0}, endt[8] = {0}; int cnt[8] = {0}; long rounds_x[8] = {0};
        for (int b = 0; b < G; ++b) {
            const int x = (int)r[b].xcc;
            clk[x] += (double)r[b].cycles / ((double)(r[b].t1 - r[b].t0) * 10.0);      // cycles per ns -> GHz
            endt[x] = std::max(endt[x], (double)(r[b].t1 - tmin) * 0.01);              // us
            cnt[x]++; rounds_x[x] += n[b];
        }
        long total = 0; for (int b = 0; b < G; ++b) total += n[b];
        printf("phase %d: %.4f ms per launch (host events over %d launches), last launch %.1f us; total rounds %ld\n", phase, ms / (L / 2), L / 2,
               (double)(tmax - tmin) * 0.01, total);
        double mean = 0;
        for (int x = 0; x < 8; ++x) { clk[x] /= std::max(cnt[x], 1); mean += clk[x] / 8; }
        for (int x = 0; x < 8; ++x)
            printf("   XCD %d: %3d workgroups  %.0f MHz (%.3f of mean)  rounds/wg %.0f  last end %.1f us\n", x, cnt[x], clk[x] * 1000, clk[x] / mean,
                   (double)rounds_x[x] / std::max(cnt[x], 1), endt[x]);
        // next phase: shares in proportion to the clock just measured (phase 0 -> 1), refined once more (1 -> 2), then back to equal shares (3)
        // to see the drift of the box itself
        if (phase < 2) {
            for (int x = 0; x < 8; ++x) w[x] *= 1.0;      // (weights are re-derived from the clocks, not accumulated)
            double tot = 0; std::vector<double> share(G);
            for (int b = 0; b < G; ++b) { share[b] = clk[(int)r[b].xcc]; tot += share[b]; }
            for (int b = 0; b < G; ++b) n[b] = (int)((double)N * G * share[b] / tot + 0.5);
        } else {
            std::fill(n.begin(), n.end(), N);
        }
    }
    return 0;
}
