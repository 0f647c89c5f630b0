#!/usr/bin/env python3
"""Wall time of the DROP-IN call -- model.predict(txt_loader, vis_loader, 'cosine') as predictor.py:197 makes it -- at C4, fed by
loaders shaped like the reference's (batch size 64, one dict per batch, shell/retrieval_task.sh:161), features already on the
device.  Three figures: the reference-shaped loop (one tower launch set per batch), the coalesced route (retrieve()'s default: all
batches collected, one launch set per tower) and retrieve() alone (scores stay in HBM; predict() adds the 1.6 GB copy to the host).

    python tools/predict_wall.py [workload] [batch]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from laff_amd import synth  # noqa: E402
import laff_amd.model.model as M  # noqa: E402


class _DS:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


class VisLoader:
    """(the id strings exist before the loop starts, as in a Dataset: making them is not part of what is timed)"""

    def __init__(self, feats, bs):
        self.feats, self.bs = feats, bs
        self.n = next(iter(feats.values())).shape[0]
        self.dataset = _DS(self.n)
        self.ids = ['video%d' % i for i in range(self.n)]

    def __iter__(self):
        for s in range(0, self.n, self.bs):
            e = min(self.n, s + self.bs)
            yield {'vis_feat_dict': {k: v[s:e] for k, v in self.feats.items()}, 'idxs': list(range(s, e)),
                   'vis_ids': tuple(self.ids[s:e]), 'vis_frame_feat_dict': {}}


class TxtLoader:
    def __init__(self, feats, gt, bs):
        self.feats, self.gt, self.bs = feats, gt, bs
        self.n = next(iter(feats.values())).shape[0]
        self.dataset = _DS(self.n)
        self.ids = ['video%d#%d' % (gt[i], i) for i in range(self.n)]

    def __iter__(self):
        for s in range(0, self.n, self.bs):
            e = min(self.n, s + self.bs)
            ids = tuple(self.ids[s:e])
            cap = {'caption': list(ids)}
            cap.update({k: v[s:e] for k, v in self.feats.items()})
            yield cap, list(range(s, e)), ids


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c4_40kx10k'
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    Nt, Nv, H, d, frames = synth.WORKLOADS[name]
    dev = torch.device('cuda:0')
    M.FC_PRECISION = 'fp16x3'
    model = synth.build_model(H, d, dev, frames=frames)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, frames=frames)
    vl, tl = VisLoader(vis, bs), TxtLoader(txt, gt.cpu().tolist(), bs)
    model.sim_precision = 'fp16'

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            t.append(time.perf_counter() - t0)
        return min(t) * 1e3

    model.coalesce_loader_batches = False
    a = timed(lambda: model.retrieve(tl, vl), reps=2)
    model.coalesce_loader_batches = True
    b = timed(lambda: model.retrieve(tl, vl))
    c = timed(lambda: model.predict(tl, vl, 'cosine'))
    if os.environ.get('PREDICT_WALL_PROFILE'):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        model.retrieve(tl, vl)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
    print('%s, loader batches of %d (%d + %d batches), fp16 similarity operands, exact ranks:' % (name, bs, (Nt + bs - 1) // bs, (Nv + bs - 1) // bs))
    print('retrieve(), one launch set per batch (the reference loop shape)  %9.2f ms  %.3e pairs/s' % (a, Nt * Nv / a * 1e3))
    print('retrieve(), batches coalesced (default)                          %9.2f ms  %.3e pairs/s' % (b, Nt * Nv / b * 1e3))
    print('predict() = the same + %d MB of scores to the host              %9.2f ms  %.3e pairs/s' % (Nt * Nv * 4 // 10**6, c, Nt * Nv / c * 1e3))


if __name__ == '__main__':
    main()
