"""A few launches of the banded similarity GEMM at C4 size for counter passes:  python tools/debug/run_strip_once.py fp16 MODE [scores]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
os.environ['LAFF_STRIP'] = sys.argv[2] if len(sys.argv) > 2 else '1'
scores = len(sys.argv) > 3 and sys.argv[3] == 'scores'
from laff_amd import ops, retrieval, synth  # noqa: E402

dev = torch.device('cuda')
Nt, Nv, K = 40000, 10000, 512
m = synth.build_model(1, 512, dev, seed=1237)
vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, seed=1237)
with torch.no_grad():
    v, t = retrieval.embed(m, vis, txt)
t, v = t.reshape(Nt, 1, K).contiguous(), v.reshape(Nv, 1, K).contiguous()
T, V = ops.pack_rows(t, True, 1e-13, prec), ops.pack_rows(v, True, 1e-13, prec)
st = ops.rank_prepare(t, v, T, V, gt)
S = torch.empty(Nt, Nv, device=dev) if scores else None
for _ in range(int(os.environ.get('REPS', '10'))):
    ops.sim_gemm_banded(st, scores, out=S)
torch.cuda.synchronize()
