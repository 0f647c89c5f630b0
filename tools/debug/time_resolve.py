"""laff_rank_resolve alone on a synthetic workload's own list (default C4): python tools/debug/time_resolve.py [workload]"""
import sys, torch
sys.path.insert(0, '.')
from laff_amd import ops, synth, retrieval
wl = sys.argv[1] if len(sys.argv) > 1 else 'c4_40kx10k'
dev = torch.device('cuda:0')
Nt, Nv, heads, d, frames = synth.WORKLOADS[wl]
spec = synth.SPECS.get(wl)
model = synth.build_model(heads, d, dev, frames=frames, seed=1237, spec=spec)
vis, txt, gt, lens = synth.make_features(Nt, Nv, dev, frames=frames, seed=1237, spec=spec)
with torch.no_grad():
    ve, te = retrieval.embed(model, vis, txt)
T, V = ops.pack_rows(te, True, 1e-13, 'fp16'), ops.pack_rows(ve, True, 1e-13, 'fp16')
def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
S = ops.alloc_scores(Nt, Nv, dev)
st = ops.rank_prepare(te, ve, T, V, gt)
st.pairs[:4].zero_()
ops.sim_gemm_banded(st, out=S)
print(wl, 'pairs', st.listed_pairs(), 'header', st.pairs[:4].tolist())
c0 = st.count.clone()
ops.rank_resolve(st, S)
print('counts changed by resolve on %d rows' % int((st.count != c0).sum()))
print('resolve        %.4f ms' % timeit(lambda: ops.rank_resolve(st, S)))
print('resolve no S   %.4f ms' % timeit(lambda: ops.rank_resolve(st, None)))
# the two-kernel form: export the listed pairs as one plain bucket (laff_rank_export_pairs, world = 1), then laff_rank_resolve on it
st = ops.rank_prepare(te, ve, T, V, gt)
st.pairs[:4].zero_()
ops.sim_gemm_banded(st, out=S)
bounds = torch.tensor([0, Nt], dtype=torch.int32, device=dev)
cap = 1 << 17
def two():
    out, fill = ops.rank_export_pairs(st, S, bounds, 0, cap)
    lst = torch.empty(4 + 2 * cap, dtype=torch.int32, device=dev)
    lst[:4] = torch.tensor([0, 0, cap, 4], dtype=torch.int32, device=dev)
    lst[4:] = out.reshape(-1)
    return ops.rank_resolve_list(te, ve, st.s_gt64, st.count, lst), fill
c, fill = two(); torch.cuda.synchronize()
print('export fill', fill.tolist())
out, fill = ops.rank_export_pairs(st, S, bounds, 0, cap)
lst = torch.empty(4 + 2 * cap, dtype=torch.int32, device=dev)
lst[:4] = torch.tensor([0, 0, cap, 4], dtype=torch.int32, device=dev)
lst[4:] = out.reshape(-1)
print('export alone        %.4f ms' % timeit(lambda: ops.rank_export_pairs(st, S, bounds, 0, cap)))
print('resolve_list alone  %.4f ms' % timeit(lambda: ops.rank_resolve_list(te, ve, st.s_gt64, st.count, lst)))
