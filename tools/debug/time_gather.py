"""fuse launch time of the C1 text side as a function of the caption length (rows gathered per item)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev = 'cuda'
torch.manual_seed(0)
N, H, d, Dk = 59800, 8, 512, 7811
wt = torch.randn(Dk, H * d, device=dev) / 30
bias = torch.randn(H * d, device=dev) * 0.1
sc = torch.rand(H * d, device=dev) + 0.5; sh = torch.randn(H * d, device=dev) * 0.1
clip = torch.randn(N, d, device=dev)
w = torch.randn(H, d, device=dev) * 0.2; b = torch.randn(H, device=dev) * 0.1; gw = torch.full((H,), 0.6, device=dev)
flags = ops.attention_flags(with_ave=True)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for k in (1, 4, 8, 12, 16, 24):
    crow = (torch.arange(N + 1, device=dev) * k).int()
    col = torch.clamp((Dk ** torch.rand(N * k, device=dev) - 1.0).long(), 0, Dk - 1).int()
    csr = torch.sparse_csr_tensor(crow, col, torch.ones(N * k, device=dev), size=(N, Dk))
    t0 = timeit(lambda: ops.fuse([(None, False, sc, sh, 'tanh', (csr, wt, bias)), (clip, True, sc, sh)], H, d, w, b, gw, flags, packed_precision='fp16'))
    print('rows per caption %2d: %.4f ms' % (k, t0))
# the alternative: a separate gather-FC launch writing the projected plane, then a fuse over two ordinary planes
k = 14
crow = (torch.arange(N + 1, device=dev) * k).int()
col = torch.clamp((Dk ** torch.rand(N * k, device=dev) - 1.0).long(), 0, Dk - 1).int()
csr = torch.sparse_csr_tensor(crow, col, torch.ones(N * k, device=dev), size=(N, Dk))
t_g = timeit(lambda: ops.fc_gather_act_bn(csr, wt, bias, sc, sh, 'tanh'))
y = ops.fc_gather_act_bn(csr, wt, bias, sc, sh, 'tanh')
t_f = timeit(lambda: ops.fuse([(y, False, None, None), (clip, True, sc, sh)], H, d, w, b, gw, flags, packed_precision='fp16'))
t_i = timeit(lambda: ops.fuse([(None, False, sc, sh, 'tanh', (csr, wt, bias)), (clip, True, sc, sh)], H, d, w, b, gw, flags, packed_precision='fp16'))
print('14 rows per caption: separate gather FC %.4f ms + fuse of two dense planes %.4f ms = %.4f ms;  gather inside the fuse launch %.4f ms' % (t_g, t_f, t_g + t_f, t_i))
