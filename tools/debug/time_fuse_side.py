"""The two fuse launches of a C4-shaped pass, with and without laff_rank_prepare's work riding in them."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from laff_amd import ops
dev = torch.device('cuda:0')
Nt, Nv, H, d, L = 40000, 10000, 1, 512, 4
g = torch.Generator(device=dev); g.manual_seed(3)
def planes(N):
    return [(torch.tanh(torch.randn(N, H * d, device=dev, generator=g)), False, None, None) for _ in range(L)]
pt, pv = planes(Nt), planes(Nv)
w = torch.randn(H, d, device=dev, generator=g) * 0.05
b = torch.zeros(H, device=dev); gw = torch.zeros(H, device=dev)
flags = ops.attention_flags()
gt = (torch.arange(Nt, device=dev) % Nv).to(torch.int32)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
print('video plain       %.1f us' % timeit(lambda: ops.fuse(pv, H, d, w, b, gw, flags)))
print('video + packed    %.1f us' % timeit(lambda: ops.fuse(pv, H, d, w, b, gw, flags, packed_precision='fp16')))
print('text plain        %.1f us' % timeit(lambda: ops.fuse(pt, H, d, w, b, gw, flags)))
print('text + packed     %.1f us' % timeit(lambda: ops.fuse(pt, H, d, w, b, gw, flags, packed_precision='fp16')))
def both():
    fp = ops.FusedPrepare(Nt, Nv, gt, heads=H)
    ops.fuse(pv, H, d, w, b, gw, flags, packed_precision='fp16', rank_side=fp.video)
    ops.fuse(pt, H, d, w, b, gw, flags, packed_precision='fp16', rank_side=fp.text)
print('video + text with the rank side (incl. allocations)  %.1f us' % timeit(both))
fp = ops.FusedPrepare(Nt, Nv, gt, heads=H)
ops.fuse(pv, H, d, w, b, gw, flags, packed_precision='fp16', rank_side=fp.video)
print('text + rank side  %.1f us' % timeit(lambda: ops.fuse(pt, H, d, w, b, gw, flags, packed_precision='fp16', rank_side=fp.text)))
print('video + rank side %.1f us' % timeit(lambda: ops.fuse(pv, H, d, w, b, gw, flags, packed_precision='fp16', rank_side=fp.video)))
