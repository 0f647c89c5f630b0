"""Random problem sizes through the whole single-GPU pass (dist.evaluate_sharded: towers with the fused rank-prepare, banded GEMM,
resolve, metrics): ranks equal those of the float64 scores of the pass's own fp32 embeddings.   python tools/debug/fuzz_pass.py [n] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from laff_amd import synth  # noqa: E402
from laff_amd.dist import HipBackend, evaluate_sharded  # noqa: E402
import laff_amd.model.model as M  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda')
M.FC_PRECISION = 'fp16x3'
bad = 0
for case in range(n_cases):
    Nt, Nv = int(rng.integers(200, 45000)), int(rng.integers(64, 12000))
    prec = 'fp16' if rng.random() < 0.7 else 'bf16'
    seed = int(rng.integers(1 << 20))
    model = synth.build_model(1, 512, dev, seed=seed)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, seed=seed)
    be = HipBackend(model, prec)
    res = evaluate_sharded(be, vis, txt, gt, Nt, Nv, 1)
    Et, Ev = res['txt_emb'].double(), res['vis_emb'].double()
    Et = Et / (Et.pow(2).sum(-1, keepdim=True).sqrt() + (1e-13 + 1e-14))
    Ev = Ev / (Ev.pow(2).sum(-1, keepdim=True).sqrt() + (1e-13 + 1e-14))
    want = torch.empty(Nt, dtype=torch.int32, device=dev)
    for a in range(0, Nt, 4096):
        S = torch.einsum('thd,vhd->tv', Et[a:a + 4096], Ev)
        g = gt[a:a + 4096].long()
        ab = S > S.gather(1, g[:, None])
        ab[torch.arange(ab.shape[0], device=dev), g] = False
        want[a:a + 4096] = ab.sum(1).to(torch.int32) + 1
    ok = torch.equal(res['ranks'].to(torch.int32), want)
    bad += not ok
    print('%2d  %6d x %6d %s : %s  R@1 %.2f' % (case, Nt, Nv, prec, 'ok' if ok else 'RANKS DIFFER', res['metrics'][0]), flush=True)
print('failures:', bad)
sys.exit(1 if bad else 0)
