"""Launch time of laff_rank_resolve / laff_rank_prepare at C4 (bench workload's embeddings): python tools/debug/time_resolve.py [fp16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from laff_amd import ops, retrieval, synth  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
dev = torch.device('cuda')
Nt, Nv, K = 40000, 10000, 512
m = synth.build_model(1, 512, dev, seed=1237)
vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, seed=1237)
with torch.no_grad():
    v, t = retrieval.embed(m, vis, txt)
t, v = t.reshape(Nt, 1, K).contiguous(), v.reshape(Nv, 1, K).contiguous()
T, V = ops.pack_rows(t, True, 1e-13, prec), ops.pack_rows(v, True, 1e-13, prec)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(best)[2]


st = ops.rank_prepare(t, v, T, V, gt)
for scores in (False, True):
    S = ops.sim_gemm_banded(st, scores)
    print('%s scores=%s: %d pairs listed; rank_resolve %.1f us, rank_prepare %.1f us' % (
        prec, scores, st.listed_pairs()[0], timed(lambda: ops.rank_resolve(st, S)), timed(lambda: ops.rank_prepare(t, v, T, V, gt))))
