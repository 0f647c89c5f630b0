"""Per-workgroup cycle stamps of the strip GEMM (debug build: tools/debug/build_trace.sh -> scratch/tracelib, loaded through
LAFF_LIB).   python tools/debug/trace_strip.py [fp16|bf16] [mode]     mode: 2 serial (default), 3 K loops only, 1 production"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
mode = sys.argv[2] if len(sys.argv) > 2 else '2'
scores = len(sys.argv) > 3 and sys.argv[3] == 'scores'
os.environ['LAFF_STRIP'] = mode
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
for _ in range(3):
    ops.sim_gemm_banded(st, scores, out=S)
torch.cuda.synchronize()
nwg = 256
tr = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
os.environ['LAFF_GEMM_TRACE_PTR'] = str(tr.data_ptr())
ops.sim_gemm_banded(st, scores, out=S)
torch.cuda.synchronize()
os.environ.pop('LAFF_GEMM_TRACE_PTR')
a = tr.cpu().numpy().reshape(nwg, 64)
t0 = a[:, 0].min()
print('precision', prec, 'mode', mode, 'scores', scores)
print('kernel span (cycles): %d   starts spread %d' % (max(a[:, 2 + 20 * s + 16].max() for s in range(3)) - t0, a[:, 0].max() - t0))
print('table staged: %.0f' % (a[:, 1] - a[:, 0]).mean())
for s in range(3):
    b = 2 + 20 * s
    live = a[:, b + 16] > 0
    if not live.any():
        continue
    x = a[live]
    n = x[:, b + 17]
    print('segment %d: %d workgroups, blocks mean %.1f (min %d max %d)' % (s, live.sum(), n.mean(), n.min(), n.max()))
    print('   strip loads issued %7.0f   prologue landed + barrier %7.0f' % ((x[:, b + 1] - x[:, b]).mean(), (x[:, b + 2] - x[:, b + 1]).mean()))
    ks = []
    for k in range(12):
        ok = n > k
        if ok.any():
            prev = x[ok, b + 2] if k == 0 else x[ok, b + 3 + k - 1]
            if k == 1:
                prev = x[ok, b + 15]
            ks.append((k, (x[ok, b + 3 + k] - prev).mean(), np.percentile(x[ok, b + 3 + k] - prev, 90)))
    print('   K loop of block k (k >= 2 includes the epilogue of block k - 1): ' + '  '.join('%d:%.0f/p90 %.0f' % z for z in ks))
    ok = n > 0
    print('   epilogue of block 0 %7.0f' % (x[ok, b + 15] - x[ok, b + 3]).mean())
    tot = x[:, b + 16] - x[:, b]
    print('   segment total %8.0f   per block %7.0f' % (tot.mean(), (tot / np.maximum(n, 1)).mean()))
busy = sum(np.where(a[:, 2 + 20 * s + 16] > 0, a[:, 2 + 20 * s + 16] - a[:, 2 + 20 * s], 0) for s in range(3))
print('workgroup busy cycles: mean %.0f  min %.0f  max %.0f ; ideal MFMA issue %.0f' % (busy.mean(), busy.min(), busy.max(), 96.29 * 4096))
