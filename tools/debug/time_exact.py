"""Per-launch timings of the exact-rank pipeline on a synthetic workload's own embeddings (default C4)."""
import sys, torch
sys.path.insert(0, '.')
from laff_amd import ops, synth, retrieval
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4_40kx10k'
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp16'
dev = torch.device('cuda:0')
Nt, Nv, heads, d, frames = synth.WORKLOADS[wl]
spec = synth.SPECS.get(wl)
model = synth.build_model(heads, d, dev, frames=frames, seed=1237, spec=spec)
vis, txt, gt, lens = synth.make_features(Nt, Nv, dev, frames=frames, seed=1237, spec=spec)
with torch.no_grad():
    ve, te = retrieval.embed(model, vis, txt)
T, V = ops.pack_rows(te, True, 1e-13, prec), ops.pack_rows(ve, True, 1e-13, prec)
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
sg = ops.row_dot_gt(T, V, gt, heads)
print(wl, prec)
print('plain S        %.4f ms' % timeit(lambda: ops.sim_gemm(T, V, heads=heads, out=S)))
print('legacy count   %.4f ms' % timeit(lambda: ops.sim_gemm(T, V, heads=heads, out=S, gt_col=gt, s_gt=sg, count=cnt)))
print('legacy no S    %.4f ms' % timeit(lambda: ops.sim_gemm(T, V, heads=heads, want_scores=False, gt_col=gt, s_gt=sg, count=cnt)))
print('row_dot_gt     %.4f ms' % timeit(lambda: ops.row_dot_gt(T, V, gt, heads)))
st = ops.rank_prepare(te, ve, T, V, gt)
print('prepare        %.4f ms' % timeit(lambda: ops.rank_prepare(te, ve, T, V, gt)))
def banded():
    st.pairs[:4].zero_()
    ops.sim_gemm_banded(st, out=S)
print('banded (+fill) %.4f ms' % timeit(banded), 'pairs', st.listed_pairs())
def banded_nos():
    st.pairs[:4].zero_()
    ops.sim_gemm_banded(st, want_scores=False)
print('banded no S    %.4f ms' % timeit(banded_nos))
print('fill only      %.4f ms' % timeit(lambda: st.pairs[:4].zero_()))
banded()
print('resolve        %.4f ms' % timeit(lambda: ops.rank_resolve(st, S)))
print('resolve no S   %.4f ms' % timeit(lambda: ops.rank_resolve(st, None)))
# where does the banded epilogue's extra time go?  (a) no in-band pairs (band = 0): only the ground-truth blocks take the slow path;
# (b) additionally no ground truth in any tile: pure fast path
st2 = ops.rank_prepare(te, ve, T, V, gt)
st2.band_t.zero_(); st2.band_v.zero_()
print('band=0            S %.4f  noS %.4f' % (timeit(lambda: ops.sim_gemm_banded(st2, out=S)), timeit(lambda: ops.sim_gemm_banded(st2, want_scores=False))))
st2.gt_col = torch.full_like(gt, -100000)
print('band=0, no gt     S %.4f  noS %.4f' % (timeit(lambda: ops.sim_gemm_banded(st2, out=S)), timeit(lambda: ops.sim_gemm_banded(st2, want_scores=False))))
gtn = torch.full_like(gt, -100000)
print('legacy, no gt     S %.4f  noS %.4f' % (timeit(lambda: ops.sim_gemm(T, V, heads=heads, out=S, gt_col=gtn, s_gt=sg, count=cnt)), timeit(lambda: ops.sim_gemm(T, V, heads=heads, want_scores=False, gt_col=gtn, s_gt=sg, count=cnt))))
st3 = ops.rank_prepare(te, ve, T, V, gt)
st3.band_t.zero_(); st3.band_v.zero_()
ops.sim_gemm_banded(st3, want_scores=False)
print('resolve, empty list (scan only) %.4f ms' % timeit(lambda: ops.rank_resolve(st3, None)), st3.listed_pairs(), 'header', st3.pairs[:4].tolist())
