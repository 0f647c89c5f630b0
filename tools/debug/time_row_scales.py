"""laff_row_scales_grouped on the C4 input set (8 matrices, 410 MB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C, torch
from laff_amd import ops
dev = 'cuda'
rows = [40000] * 4 + [10000] * 4
X = [torch.randn(n, 512, device=dev) for n in rows]
R = [torch.empty(n, device=dev) for n in rows]
lib, h = ops._context(torch.device(dev))
n = len(rows)
Xp, N, K, LD, Rp = (C.c_void_p * n)(), (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)(), (C.c_void_p * n)()
for i in range(n):
    Xp[i], N[i], K[i], LD[i], Rp[i] = X[i].data_ptr(), rows[i], 512, 512, R[i].data_ptr()
def run(): ops._call('row_scales', lib.laff_row_scales_grouped, h, n, Xp, N, K, LD, Rp)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 50)
ref = [torch.exp2(torch.floor(torch.log2(x.abs().amax(1))) - 9) for x in X]
print('row_scales %.4f ms  %.2f TB/s  correct %s' % (best, sum(rows) * 2048 / best / 1e9, all(torch.equal(a, b) for a, b in zip(R, ref))))
