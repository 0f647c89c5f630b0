"""Launch time of the banded similarity GEMM at C4 (bench workload's embeddings): python tools/debug/time_strip.py fp16 MODE [scores [ldo]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
os.environ['LAFF_STRIP'] = sys.argv[2] if len(sys.argv) > 2 else '1'
scores = len(sys.argv) > 3 and sys.argv[3] == 'scores'
from laff_amd import ops, retrieval, synth  # noqa: E402

dev = torch.device('cuda')
Nt, Nv, K = int(os.environ.get('NT', 40000)), int(os.environ.get('NV', 10000)), 512
m = synth.build_model(1, 512, dev, seed=1237)
vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, seed=1237)
with torch.no_grad():
    v, t = retrieval.embed(m, vis, txt)
t, v = t.reshape(Nt, 1, K).contiguous(), v.reshape(Nv, 1, K).contiguous()
ps = float(sys.argv[5]) if len(sys.argv) > 5 else None
T, V = ops.pack_rows(t, True, 1e-13, prec, ps), ops.pack_rows(v, True, 1e-13, prec, ps)
st = ops.rank_prepare(t, v, T, V, gt)
ldo = int(sys.argv[4]) if len(sys.argv) > 4 else Nv
S = torch.empty(Nt, ldo, device=dev)[:, :Nv] if scores else None
for _ in range(5):
    ops.sim_gemm_banded(st, scores, out=S)
torch.cuda.synchronize()
best = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.sim_gemm_banded(st, scores, out=S)
    e1.record()
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 20)
print('%dx%d' % (Nt, Nv), '%s ldo %d LAFF_STRIP=%s scores=%s lib=%s : %.4f ms (min of 5 x 20: %.4f)' % (prec, ldo, os.environ['LAFF_STRIP'], scores, os.environ.get('LAFF_HIP_LIB', 'default'), sorted(best)[2], min(best)))
