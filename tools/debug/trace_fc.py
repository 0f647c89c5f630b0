"""Per-tile timeline of the fused-split FC launch at C4 (debug library with LAFF_GEMM_TRACE: tools/debug/build_trace.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from laff_amd import ops
dev = 'cuda'
torch.manual_seed(0)
rows = [40000] * 4 + [10000] * 4
W = [torch.randn(512, 512, device=dev) / 22 for _ in rows]
Ws = [ops.split_rows(w) for w in W]
X = [torch.randn(n, 512, device=dev) for n in rows]
b = torch.randn(512, device=dev) * 0.1
sc = torch.rand(512, device=dev) + 0.5
sh = torch.randn(512, device=dev) * 0.1
act = sys.argv[1] if len(sys.argv) > 1 else 'tanh'
probs = [dict(x=X[i], weight_split=Ws[i], bias=b, bn_scale=sc if act != 'none' else None, bn_shift=sh if act != 'none' else None,
              activation=None if act == 'none' else act) for i in range(8)]
for _ in range(3): ops.fc_act_bn_fused_grouped(probs)
torch.cuda.synchronize()
a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a0.record()
for _ in range(10): ops.fc_act_bn_fused_grouped(probs)
b0.record(); torch.cuda.synchronize()
print('fused FC (row_scales + GEMM) %.4f ms per call, act=%s' % (a0.elapsed_time(b0) / 10, act))
nb = sum(((n + 255) // 256) * 2 for n in rows)
tr = torch.zeros(nb * 8, dtype=torch.int64, device=dev)
os.environ['LAFF_GEMM_TRACE_PTR'] = str(tr.data_ptr())
ops.fc_act_bn_fused_grouped(probs); torch.cuda.synchronize()
os.environ.pop('LAFF_GEMM_TRACE_PTR')
a = tr.cpu().numpy().reshape(nb, 8)
d = np.diff(a[:, :7], axis=1)
for i, n in enumerate(['setup', 'prologue', 'kstep0', 'ksteps 1..15', 'barrier', 'epilogue']):
    print('%-14s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f' % (n, d[:, i].mean(), *np.percentile(d[:, i], [10, 50, 90])))
tot = a[:, 6] - a[:, 0]
print('WG total       mean %8.0f  p50 %8.0f   (MFMA issue alone: 16 K-steps x 48 MFMA x 32 cycles x 2 waves per SIMD = 49152)' % (tot.mean(), np.median(tot)))
ids = a[:, 7]; key = ((ids >> 32) & 0xf) * 100000 + (ids & 0xffff00)
gaps = []
for k in np.unique(key):
    m = key == k; st = np.sort(a[m, 0]); en = a[m, 6][np.argsort(a[m, 0])]
    gaps.extend((st[1:] - en[:-1]).tolist())
print('tiles', nb, 'distinct CUs', len(np.unique(key)), 'gap between tiles on a CU: mean %.0f p50 %.0f' % (np.mean(gaps), np.median(gaps)))
span = a[:, 6].max() - a[:, 0].min()
cnt = np.array([np.sum(key == k) for k in np.unique(key)])
print('launch span %d cycles; tiles per CU: min %d max %d; busy fraction of the span (sum of WG totals / CUs / span): %.3f' % (
    span, cnt.min(), cnt.max(), tot.sum() / len(cnt) / span))
# when does each CU finish its last tile, relative to the span
last = np.array([a[key == k, 6].max() for k in np.unique(key)]) - a[:, 0].min()
print('CU finish time / span: p10 %.2f p50 %.2f p90 %.2f' % tuple(np.percentile(last / span, [10, 50, 90])))
first = np.array([a[key == k, 0].min() for k in np.unique(key)]) - a[:, 0].min()
print('CU first start (cycles): p50 %.0f p90 %.0f max %.0f' % (np.median(first), np.percentile(first, 90), first.max()))
