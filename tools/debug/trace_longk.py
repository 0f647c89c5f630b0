"""Per-K-step wait breakdown of wave 0 in laff_sim_gemm at long K (debug library with LAFF_GEMM_TRACE: tools/debug/build_trace.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from laff_amd import ops
dev = 'cuda'; N = int(sys.argv[3]) if len(sys.argv) > 3 else 8192; K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
t = torch.nn.functional.normalize(torch.randn(N, K, device=dev), dim=1); v = torch.nn.functional.normalize(torch.randn(N, K, device=dev), dim=1)
T = ops.pack_rows(t, True, 1e-13, 'fp16'); V = ops.pack_rows(v, True, 1e-13, 'fp16')
S = torch.empty(N, N, device=dev)
gt = (torch.arange(N, device=dev) % N).int()
t = torch.nn.functional.normalize(t + 0.5 * v[gt.long()], dim=1)        # a planted match: a realistic share of pairs inside the band
T = ops.pack_rows(t, True, 1e-13, 'fp16')
st = ops.rank_prepare(t, v, T, V, gt)
banded = len(sys.argv) > 2 and sys.argv[2] == 'banded'
def run():
    if banded: ops.sim_gemm_banded(st, want_scores=True, out=S)
    else: ops.sim_gemm(T, V, out=S)
for _ in range(3): run()
torch.cuda.synchronize()
nb = (N // 256) ** 2
tr = torch.zeros(nb * 24, dtype=torch.int64, device=dev)
os.environ['LAFF_GEMM_TRACE_PTR'] = str(tr.data_ptr())
run(); torch.cuda.synchronize()
os.environ.pop('LAFF_GEMM_TRACE_PTR')
full = tr.cpu().numpy()
a = full[:nb * 8].reshape(nb, 8); w = full[nb * 8:nb * 16].reshape(nb, 8)
d = np.diff(a[:, :7], axis=1)
for i, n in enumerate(['setup', 'prologue', 'kstep0', 'ksteps 1..', 'barrier', 'epilogue']):
    print('%-12s mean %9.0f p50 %9.0f' % (n, d[:, i].mean(), np.median(d[:, i])))
nkt = w[:, 3].astype(float)
loop = (w[:, 7] - w[:, 6]).astype(float)
print('K loop: %.0f cycles per K-step (MFMA issue alone %d); waits per K-step: LDS fragments %.0f, DMA landing %.0f, barrier %.0f' % (
    (loop / nkt).mean(), 2048, (w[:, 0] / nkt).mean(), (w[:, 1] / nkt).mean(), (w[:, 2] / nkt).mean()))
if banded:
    e = full[nb * 16:nb * 24].reshape(nb, 8)
    de = np.diff(e, axis=1)
    for i, n in enumerate(['setup', 'rowblk0', 'rowblk1', 'rowblk2', 'rowblk3', 'publish', 'overflow']):
        print('  banded epilogue %-10s mean %8.0f p50 %8.0f' % (n, de[:, i].mean(), np.median(de[:, i])))
