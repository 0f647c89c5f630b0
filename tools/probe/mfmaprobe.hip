// lone wave per SIMD: cycles per v_mfma_f32_32x32x16_f16 with K filler instructions behind each one
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int K, int KIND, bool AGPR>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, unsigned* sink, int iters) {
    extern __shared__ char smem[];
    f32x16 acc[4];
    u32x4 a = {threadIdx.x, 1, 2, 3}, b = {threadIdx.x * 3u, 5, 6, 7};
    unsigned f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
    u32x4 ld[4];
    const unsigned laddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    asm volatile("" : "+v"(a), "+v"(b));
    u32x4 ba = b;
    if (AGPR) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ba.x) : "v"(b.x));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a), "a"(ba));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]));          // independent VALU
                else if (KIND == 1) asm volatile("v_add_u32 %0, %0, 1" : "+v"(f[0]));                                    // dependent chain
                else if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[k & 3]) : "v"(laddr));                // LDS reads
                else if (KIND == 3) asm volatile("s_nop 0");
                else if (KIND == 4) asm volatile("v_sub_f32 %0, %1, %2\n\tv_alignbit_b32 %3, %3, %0, 31" : "=&v"(f[k & 3]), "+v"(f[4 + (k & 3)]) : "v"(f[7]), "v"(acc[(m + 1) % NACC][k]) : );
            }
            if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(8)");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 4; ++i) s += ld[i].x + (unsigned)acc[i][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int K, int KIND, bool AGPR>
void run(const char* name, unsigned long long* d, unsigned* sink) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)probe<NACC, K, KIND, AGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<NACC, K, KIND, AGPR>), dim3(256), dim3(256), 100 * 1024, 0, d, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-34s acc %d  fillers %d  agprB %d : %.1f cycles / MFMA\n", name, NACC, K, (int)AGPR, (double)h / (iters * 16.0));
}
int main() {
    unsigned long long* d; unsigned* sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 256 * 256 * 4);
    run<4, 0, 0, false>("bare", d, sink);
    run<2, 0, 0, false>("bare", d, sink);
    run<1, 0, 0, false>("bare", d, sink);
    run<4, 0, 0, true>("bare", d, sink);
    run<2, 0, 0, true>("bare", d, sink);
    run<4, 2, 0, true>("indep VALU", d, sink);
    run<4, 4, 0, true>("indep VALU", d, sink);
    run<4, 6, 0, true>("indep VALU", d, sink);
    run<2, 4, 0, true>("indep VALU", d, sink);
    run<4, 2, 1, true>("dependent VALU chain", d, sink);
    run<4, 4, 1, true>("dependent VALU chain", d, sink);
    run<4, 1, 2, true>("ds_read_b128", d, sink);
    run<4, 2, 2, true>("ds_read_b128", d, sink);
    run<4, 4, 3, true>("s_nop 0", d, sink);
    run<4, 2, 4, true>("sub+alignbit pairs (reads acc)", d, sink);
    run<2, 2, 4, true>("sub+alignbit pairs (reads acc)", d, sink);
    run<4, 3, 4, true>("sub+alignbit pairs (reads acc)", d, sink);
    // one accumulator (every MFMA depends on the one before it: fc_strip's chain of three per sub-step)
    run<1, 0, 0, true>("bare, ONE accumulator", d, sink);
    run<1, 1, 3, true>("s_nop 0, one accumulator", d, sink);
    run<1, 2, 3, true>("s_nop 0, one accumulator", d, sink);
    run<1, 1, 0, true>("indep VALU, one accumulator", d, sink);
    run<1, 2, 0, true>("indep VALU, one accumulator", d, sink);
    run<1, 3, 0, true>("indep VALU, one accumulator", d, sink);
    run<1, 4, 0, true>("indep VALU, one accumulator", d, sink);
    run<1, 6, 0, true>("indep VALU, one accumulator", d, sink);
    run<1, 8, 0, true>("indep VALU, one accumulator", d, sink);
    run<2, 1, 3, true>("s_nop 0", d, sink);
    run<2, 6, 0, true>("indep VALU", d, sink);
    run<2, 8, 0, true>("indep VALU", d, sink);
    return 0;
}
