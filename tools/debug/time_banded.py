import sys, torch, numpy as np
sys.path.insert(0, '.')
from laff_amd import ops
dev = 'cuda'
Nt, Nv, K = 40000, 10000, 512
g = torch.Generator(device=dev); g.manual_seed(1)
z = torch.randn(Nv, 64, device=dev, generator=g)
P = torch.randn(64, K, device=dev, generator=g)
gt = (torch.arange(Nt, device=dev) % Nv).to(torch.int32)
Ev = z @ P + 0.5 * torch.randn(Nv, K, device=dev, generator=g)
Et = z[gt.long()] @ P + 0.5 * torch.randn(Nt, K, device=dev, generator=g)
Ev = (Ev / Ev.norm(dim=1, keepdim=True)).contiguous(); Et = (Et / Et.norm(dim=1, keepdim=True)).contiguous()
T, V = ops.pack_rows(Et, True, 1e-13, 'fp16'), ops.pack_rows(Ev, True, 1e-13, 'fp16')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
S = torch.empty(Nt, Nv, device=dev)
cnt = torch.zeros(Nt, dtype=torch.int32, device=dev)
sg = ops.row_dot_gt(T, V, gt, 1)
print('plain S        %.3f ms' % timeit(lambda: ops.sim_gemm(T, V, out=S)))
print('legacy count   %.3f ms' % timeit(lambda: ops.sim_gemm(T, V, out=S, gt_col=gt, s_gt=sg, count=cnt)))
st = ops.rank_prepare(Et, Ev, T, V, gt)
print('prepare        %.3f ms' % timeit(lambda: ops.rank_prepare(Et, Ev, T, V, gt)))
def banded():
    st.pairs[:4].zero_()
    ops.sim_gemm_banded(st, out=S)
print('banded         %.3f ms' % timeit(banded), 'pairs', st.listed_pairs())
def banded_nos():
    st.pairs[:4].zero_()
    ops.sim_gemm_banded(st, want_scores=False)
print('banded no S    %.3f ms' % timeit(banded_nos))
print('legacy no S    %.3f ms' % timeit(lambda: ops.sim_gemm(T, V, want_scores=False, gt_col=gt, s_gt=sg, count=cnt)))
print('resolve        %.3f ms' % timeit(lambda: ops.rank_resolve(st, S)))
# band scaled up/down: how does the epilogue cost move with the number of listed pairs
for f in (0.0, 0.25, 4.0):
    st2 = ops.rank_prepare(Et, Ev, T, V, gt)
    st2.band_t.mul_(f); st2.band_v.mul_(f)
    def b2():
        st2.pairs[:4].zero_()
        ops.sim_gemm_banded(st2, out=S)
    print('banded band x%.2f  %.3f ms' % (f, timeit(b2)), st2.listed_pairs())
