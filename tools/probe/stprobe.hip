// store-path probe: what a buffer_store_dwordx4 costs a CU by address pattern, with and without MFMAs around it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// PAT 0: 16 rows x 64 B per instruction; 1: 8 rows x 128 B; 2: 4 rows x 256 B; 3: 1 KiB contiguous (row = 1 KiB chunks)
template <int PAT, int MF, bool NT>
__global__ __launch_bounds__(256, 1) void k(char* S, unsigned pitch, int nblocks, int active, unsigned long long* cyc, float* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long base = (unsigned long long)S + ((unsigned long long)blockIdx.x * 256 + wave * 64) * pitch;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)base);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) & 0xffffu;
    r.z = 0xffffffffu;
    r.w = 0x00020000u;
    unsigned vo[8];
    for (int s = 0; s < 8; ++s) {
        unsigned row, byte;
        if (PAT == 0) { row = 16 * (s & 3) + (lane >> 2); byte = (lane & 3) * 16 + 64 * (s >> 2); }
        else if (PAT == 1) { row = 8 * s + (lane >> 3); byte = (lane & 7) * 16; }
        else if (PAT == 2) { row = 4 * s + (lane >> 4); byte = (lane & 15) * 16; }      // (256 B per row and block)
        else { row = s; byte = lane * 16; }                                               // (1 KiB per row and block)
        vo[s] = row * pitch + byte;
    }
    const unsigned step = PAT == 2 ? 256 : (PAT == 3 ? 1024 : 128);
    f32x16 acc0 = {}, acc1 = {};
    f16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    u32x4 data = {1u, 2u, 3u, (unsigned)lane};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < nblocks; ++i) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            }
            if (wave < active) {
                if (NT) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" ::"v"(data), "v"(vo[s]), "s"(r) : "memory");
                else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(data), "v"(vo[s]), "s"(r) : "memory");
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) vo[s] += step;
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
    if (acc0[0] + acc1[0] == 123.f) sink[0] = 1.f;
}

template <int PAT, int MF, bool NT>
void run(const char* name, char* S, unsigned pitch, int active, unsigned long long* d_cyc, float* sink) {
    const int G = 256, nblocks = PAT == 3 ? 39 : (PAT == 2 ? 156 : 312);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    std::vector<unsigned long long> h(G * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<PAT, MF, NT>), dim3(G), dim3(256), 0, 0, S, pitch, nblocks, active, d_cyc, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(h.data(), d_cyc, G * 8 * 8, hipMemcpyDeviceToHost);
    double c1 = 0, c2 = 0; for (int i = 0; i < G * 4; ++i) { c1 += h[2 * i]; c2 += h[2 * i + 1]; }
    c1 /= G * 4; c2 /= G * 4;
    const double bytes = (double)G * active * nblocks * 8 * 1024;
    printf("%-34s pitch %6u waves %d mfma/store %2d: %.3f ms  %.2f TB/s  cycles/block %.0f (drained %.0f)  per store-slot %.1f\n", name, pitch, active, MF, best,
           bytes / best * 1e-9, c1 / nblocks, c2 / nblocks, c1 / nblocks / 8);
}

int main() {
    char* S; unsigned long long* d_cyc; float* sink;
    const size_t bytes = (size_t)65536 * 40960 + (1 << 20);
    hipMalloc(&S, bytes); hipMalloc(&d_cyc, 256 * 8 * 8); hipMalloc(&sink, 4);
    hipMemset(S, 0, bytes);
#define RUN(P, M, N, pitch, act) run<P, M, N>(#P "," #M "," #N, S, pitch, act, d_cyc, sink)
    for (unsigned pitch : {40000u, 40960u}) {
        for (int act : {1, 2, 4}) {
            RUN(0, 0, true, pitch, act); RUN(1, 0, true, pitch, act);
            RUN(0, 8, true, pitch, act); RUN(1, 8, true, pitch, act);
            RUN(0, 12, true, pitch, act); RUN(1, 12, true, pitch, act);
        }
        RUN(0, 8, false, pitch, 4); RUN(1, 8, false, pitch, 4);
        RUN(2, 0, true, pitch, 4); RUN(2, 8, true, pitch, 4);
        RUN(3, 0, true, pitch, 4); RUN(3, 8, true, pitch, 4);
        RUN(0, 8, true, pitch, 0);
    }
    return 0;
}
