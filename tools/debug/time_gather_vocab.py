import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev = 'cuda'
torch.manual_seed(0)
N, H, d = 59800, 8, 512
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for Dk, dist_ in ((7811, 'zipf'), (7811, 'uniform'), (2000, 'uniform'), (500, 'uniform'), (64, 'uniform'), (7811, 'same')):
    wt = torch.randn(Dk, H * d, device=dev) / 30
    bias = torch.randn(H * d, device=dev) * 0.1
    sc = torch.rand(H * d, device=dev) + 0.5; sh = torch.randn(H * d, device=dev) * 0.1
    clip = torch.randn(N, d, device=dev)
    w = torch.randn(H, d, device=dev) * 0.2; b = torch.randn(H, device=dev) * 0.1; gw = torch.full((H,), 0.6, device=dev)
    flags = ops.attention_flags(with_ave=True)
    for k in (2, 14):
        crow = (torch.arange(N + 1, device=dev) * k).int()
        if dist_ == 'zipf':
            col = torch.clamp((Dk ** torch.rand(N * k, device=dev) - 1.0).long(), 0, Dk - 1).int()
        elif dist_ == 'same':
            col = torch.zeros(N * k, device=dev, dtype=torch.int32)
        else:
            col = torch.randint(0, Dk, (N * k,), device=dev, dtype=torch.int32)
        csr = torch.sparse_csr_tensor(crow, col, torch.ones(N * k, device=dev), size=(N, Dk))
        t0 = timeit(lambda: ops.fuse([(None, False, sc, sh, 'tanh', (csr, wt, bias)), (clip, True, sc, sh)], H, d, w, b, gw, flags, packed_precision='fp16'))
        print('vocab %5d %-8s rows per caption %2d: %.4f ms' % (Dk, dist_, k, t0))
y = torch.randn(N, H * d, device=dev)
t_f = timeit(lambda: ops.fuse([(y, False, None, None), (clip, True, sc, sh)], H, d, w, b, gw, flags, packed_precision='fp16'))
t_1 = timeit(lambda: ops.fuse([(clip, True, sc, sh), (clip, True, sc, sh)], H, d, w, b, gw, flags, packed_precision='fp16'))
print('dense two planes %.4f ms; two tiled no-transform planes (output traffic only) %.4f ms' % (t_f, t_1))
