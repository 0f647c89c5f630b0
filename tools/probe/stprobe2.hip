// store-path probe 2: is the fp32 score-tile store rate a per-CU limit or a memory-side one?  G workgroups (4 waves, one per SIMD)
// each store `ntiles` 256 x 256 fp32 tiles the way the long-K GEMM's epilogue does (global_store_dwordx4 nt, 2 rows x 512 B per
// instruction, 64 instructions per wave and tile), G = 8 .. 256: a per-CU limit keeps bytes/clk/CU constant, a memory-side limit
// lets it rise as fewer CUs store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256, 1) void k(float* S, long ldo, int tiles_c, int ntiles, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < ntiles; ++t) {
        const long tile = (long)blockIdx.x * ntiles + t;
        const long r0 = (tile / tiles_c) * 256 + wr * 128, c0 = (tile % tiles_c) * 256 + wc * 128;
#pragma unroll 16
        for (int j = 0; j < 64; ++j) {
            const long row = r0 + 2 * j + (lane >> 5);
            float* o = S + row * ldo + c0 + (lane & 31) * 4;
            if (NT) __builtin_nontemporal_store(v, (f32x4*)o);
            else *(f32x4*)o = v;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

// `stprobe2 energy [seconds]`: the G = 256 nt case in a loop, for a joules-per-byte figure (read rocm-smi --showenergycounter around it)
static int energy_loop(double seconds, int G, int nt) {
    const long N = 16384;
    float* S; unsigned long long* d_cyc;
    if (hipMalloc(&S, (size_t)N * N * 4) != hipSuccess || hipMalloc(&d_cyc, 256 * 8 * 8) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int ntiles = 16;
    (void)hipEventRecord(e0);
    long launches = 0;
    float ms = 0;
    while (ms < seconds * 1e3) {
        for (int i = 0; i < 200; ++i) {
            if (nt) hipLaunchKernelGGL(k<true>, dim3(G), dim3(256), 0, 0, S, N, (int)(N / 256), ntiles, d_cyc);
            else hipLaunchKernelGGL(k<false>, dim3(G), dim3(256), 0, 0, S, N, (int)(N / 256), ntiles, d_cyc);
        }
        launches += 200;
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)launches * G * ntiles * 256 * 256 * 4;
    printf("energy loop G %d nt %d: %ld launches, %.1f ms, %.3f GB written, %.2f TB/s\n", G, nt, launches, ms, bytes * 1e-9, bytes / ms * 1e-9);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'e') return energy_loop(argc > 2 ? atof(argv[2]) : 3.0, argc > 3 ? atoi(argv[3]) : 256, argc > 4 ? atoi(argv[4]) : 1);
    const long N = 16384;
    float* S; unsigned long long* d_cyc;
    hipMalloc(&S, (size_t)N * N * 4); hipMalloc(&d_cyc, 256 * 8 * 8);
    hipMemset(S, 0, (size_t)N * N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
        for (int G : {8, 16, 32, 64, 128, 256}) {
            const int ntiles = 16;
            float best = 1e9;
            std::vector<unsigned long long> h(G * 8);
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(k<true>, dim3(G), dim3(256), 0, 0, S, N, (int)(N / 256), ntiles, d_cyc);
                else hipLaunchKernelGGL(k<false>, dim3(G), dim3(256), 0, 0, S, N, (int)(N / 256), ntiles, d_cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            hipMemcpy(h.data(), d_cyc, G * 8 * 8, hipMemcpyDeviceToHost);
            double c1 = 0, c2 = 0; for (int i = 0; i < G * 4; ++i) { c1 += h[2 * i]; c2 += h[2 * i + 1]; }
            c1 /= G * 4; c2 /= G * 4;
            const double bytes = (double)G * ntiles * 256 * 256 * 4;
            printf("%s G %3d: %.3f ms  %.2f TB/s  issue %.0f cycles/tile, drained %.0f cycles/tile = %.1f B/clk/CU\n", nt ? "nt   " : "plain", G, best,
                   bytes / best * 1e-9, c1 / ntiles, c2 / ntiles, 262144.0 / (c2 / ntiles));
        }
    return 0;
}
