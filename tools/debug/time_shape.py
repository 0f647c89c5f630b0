"""laff_sim_gemm (plain and banded) at one shape: python tools/debug/time_shape.py Nt Nv K [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev = 'cuda'
Nt, Nv, K = (int(x) for x in sys.argv[1:4]); prec = sys.argv[4] if len(sys.argv) > 4 else 'fp16'
torch.manual_seed(0)
t = torch.nn.functional.normalize(torch.randn(Nt, K, device=dev), dim=1); v = torch.nn.functional.normalize(torch.randn(Nv, K, device=dev), dim=1)
gt = (torch.arange(Nt, device=dev) % Nv).int()
t = torch.nn.functional.normalize(t + 0.5 * v[gt.long()], dim=1)
T = ops.pack_rows(t, True, 1e-13, prec); V = ops.pack_rows(v, True, 1e-13, prec)
S = torch.empty(Nt, Nv, device=dev)
st = ops.rank_prepare(t, v, T, V, gt)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
fl = 2.0 * Nt * Nv * K / 1e9
a = timeit(lambda: ops.sim_gemm(T, V, out=S)); b = timeit(lambda: ops.sim_gemm_banded(st, want_scores=True, out=S)); c = timeit(lambda: ops.sim_gemm_banded(st, want_scores=False))
print('%d x %d x %d %s  variant %s: plain+S %.4f ms %.0f TF | banded+S %.4f ms %.0f TF | banded count-only %.4f ms %.0f TF' % (
    Nt, Nv, K, prec, os.environ.get('LAFF_GEMM_VARIANT', 'auto'), a, fl / a, b, fl / b, c, fl / c))
