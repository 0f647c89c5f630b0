"""Does laff_fuse read planes that were just written from the Infinity Cache?  The text-side fuse (4 planes of 512 columns + packed
operand) on N rows, (a) back to back on the same inputs, (b) after a pass that rewrites the planes, (c) after 600 MB of unrelated
traffic.   python tools/debug/time_fuse_mall.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from laff_amd import ops  # noqa: E402

dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
w = torch.randn(1, 512, generator=g, device=dev) * 0.05
b = torch.zeros(1, device=dev)
gw = torch.ones(1, device=dev)
flush = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
for N in (10000, 20000, 40000):
    planes = [torch.randn(N, 512, generator=g, device=dev) for _ in range(4)]
    pl = [(p, False, None, None) for p in planes]

    def fuse():
        return ops.fuse(pl, 1, 512, w, b, gw, 0, packed_precision='fp16')

    def timed(pre):
        ts = []
        for _ in range(8):
            pre()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fuse()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]
    fuse()
    nbytes = N * 512 * 4 * 4 + N * 512 * 4 + N * 512 * 2
    a = timed(lambda: None)
    c = timed(lambda: [p.mul_(1.0) for p in planes])
    d_ = timed(lambda: flush.fill_(1))
    print('N %6d (%.0f MB in + out): back to back %.1f us (%.2f TB/s)   planes just rewritten %.1f us   after 600 MB of other traffic %.1f us' % (
        N, nbytes / 1e6, a, nbytes / a / 1e6, c, d_))
