"""laff_sim_gemm fp16 16384 x 16384 x K for a range of K: the slope is the steady-state K loop, the intercept the per-tile fixed cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from laff_amd import ops
dev = 'cuda'
N = 16384
S = torch.empty(N, N, device=dev)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
Ks = (512, 1024, 2048, 4096, 8192)
ts = []
for K in Ks:
    t = torch.nn.functional.normalize(torch.randn(N, K, device=dev), dim=1); v = torch.nn.functional.normalize(torch.randn(N, K, device=dev), dim=1)
    T = ops.pack_rows(t, True, 1e-13, 'fp16'); V = ops.pack_rows(v, True, 1e-13, 'fp16')
    ms = timeit(lambda: ops.sim_gemm(T, V, out=S))
    ts.append(ms)
    a16, b16 = t.half(), v.half()
    C = torch.empty(N, N, device=dev, dtype=torch.float16)
    mv = timeit(lambda: torch.matmul(a16, b16.t(), out=C))
    print('K %5d: laff %.4f ms %6.0f TF   hipBLASLt %.4f ms %6.0f TF' % (K, ms, 2.0 * N * N * K / ms / 1e9, mv, 2.0 * N * N * K / mv / 1e9))
b, a = np.polyfit(np.array(Ks[2:], float), np.array(ts[2:]), 1)
print('steady state (slope over K >= 2048): %.0f TF; intercept %.3f ms' % (2.0 * N * N / b / 1e9, a))
