#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path at C4: features start in pinned HOST memory, the score matrix
ends in pinned HOST memory -- what `model.predict()` hands over and returns -- against the HBM-resident step bench.py times.

    python tools/pcie_inclusive.py [workload]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from laff_amd import retrieval, synth  # noqa: E402
import laff_amd.model.model as M  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c4_40kx10k'
    Nt, Nv, H, d, frames = synth.WORKLOADS[name]
    dev = torch.device('cuda:0')
    M.FC_PRECISION = 'fp16x3'
    model = synth.build_model(H, d, dev, frames=frames)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, frames=frames)
    vis_h = {k: v.cpu().pin_memory() for k, v in vis.items()}
    txt_h = {k: v.cpu().pin_memory() for k, v in txt.items()}
    S_h = torch.empty((Nt, Nv), dtype=torch.float32).pin_memory()
    nbytes_in = sum(v.numel() * 4 for v in list(vis_h.values()) + list(txt_h.values()))

    def once(h2d, d2h):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = {k: x.to(dev, non_blocking=True) for k, x in vis_h.items()} if h2d else vis
        t = {k: x.to(dev, non_blocking=True) for k, x in txt_h.items()} if h2d else txt
        res = retrieval.evaluate(model, v, t, gt, precision='fp16')
        if d2h:
            S_h.copy_(res.S, non_blocking=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for h2d, d2h, label in ((False, False, 'HBM-resident (eager launches)'), (True, False, '+ H2D of the features'),
                            (True, True, '+ H2D of the features + D2H of S')):
        for _ in range(2):
            once(h2d, d2h)
        dt = min(once(h2d, d2h) for _ in range(5))
        print('%-42s %8.3f ms  %.3e pairs/s' % (label, dt * 1e3, Nt * Nv / dt))
    print('features %.0f MB host->device, scores %.0f MB device->host' % (nbytes_in / 1e6, Nt * Nv * 4 / 1e6))


if __name__ == '__main__':
    main()
