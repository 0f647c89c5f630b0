#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* buf, unsigned nrec, unsigned* out) {
    unsigned long long b = (unsigned long long)buf;
    u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xffffu;
    r.z = __builtin_amdgcn_readfirstlane(nrec);
    r.w = 0x00020000u;
    unsigned voff = threadIdx.x * 16u;
    u32x4 v = {1000u + threadIdx.x, 2u, 3u, 4u};
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:0 nt" ::"v"(v), "v"(voff), "s"(r) : "memory");
    u32x4 l;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:0\n\ts_waitcnt vmcnt(0)" : "=v"(l) : "v"(voff + 4096u), "s"(r) : "memory");
    out[threadIdx.x] = l.x;
}
int main() {
    unsigned *d, *o;
    hipMalloc(&d, 1 << 16); hipMalloc(&o, 64 * 4);
    for (unsigned nrec : {0u, 1u, 16u, 100u, 256u, 1024u, 4096u + 48u}) {
        hipMemset(d, 0xAB, 1 << 16);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nrec, o);
        hipDeviceSynchronize();
        std::vector<unsigned> h(1 << 14), ho(64);
        hipMemcpy(h.data(), d, 1 << 16, hipMemcpyDeviceToHost);
        hipMemcpy(ho.data(), o, 256, hipMemcpyDeviceToHost);
        int written = 0, last = -1;
        for (int t = 0; t < 64; ++t) if (h[t * 4] == 1000u + t) { ++written; last = t; }
        int loaded = 0; for (int t = 0; t < 64; ++t) if (ho[t] == 0xABABABABu) ++loaded;
        printf("num_records %5u: lanes whose 16-byte store landed %2d (last lane %2d)   loads at +4096 returning data %d (others %x)\n", nrec, written, last, loaded, ho[63]);
    }
    return 0;
}
