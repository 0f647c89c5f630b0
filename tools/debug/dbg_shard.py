import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from laff_amd import ops
from test_gpu_kernels import _exact_scores_f64, _count_ranks, rnd, dev
g = rnd(77)
Nt, Nv, H, d = 900, 400, 1, 128
t = g.normal(0, 1, (Nt, H, d)).astype(np.float32)
v = (g.normal(0, 1, (Nv, H, d)) * 0.05 + g.normal(0, 1, (1, H, d))).astype(np.float32)
gt = dev(g.integers(0, Nv, Nt).astype(np.int32), torch.int32)
Et, Ev = dev(t), dev(v)
T = ops.pack_rows(Et, True, 1e-13, 'fp16')
S64 = _exact_scores_f64(Et, Ev)
want = _count_ranks(S64, gt)
sg_all = S64.gather(1, gt.long()[:, None])[:, 0]
for a, b in ((0, 130), (130, 131), (131, 400), (0, 400)):
    Evs = Ev[a:b].contiguous()
    V = ops.pack_rows(Evs, True, 1e-13, 'fp16')
    st = ops.rank_prepare(Et, Evs, T, V, gt, col0=a)
    st.s_gt64.copy_(sg_all)
    S = ops.sim_gemm_banded(st)
    c_gemm = st.count.clone()
    pr = st.pair_indices()
    n, ov = st.listed_pairs()
    plain = ops.sim_gemm(T, V)
    band_hi = st.band_t[:Nt, None] + st.band_v[None, :b - a].max()
    dl = plain - st.s_gt64.float()[:, None]
    cols = torch.arange(a, b, device='cuda')[None, :]
    isgt = cols == gt.long()[:, None]
    definite = (dl > band_hi) & ~isgt
    inband = (dl.abs() <= st.band_t[:Nt, None] + st.band_v[None, :b - a]) & ~isgt
    lm = torch.zeros_like(inband); lm[pr[:, 0], pr[:, 1]] = True
    dup = pr.shape[0] - lm.sum().item()
    print((a, b), 'listed', n, ov, 'hdr', st.pairs[:4].tolist(), 'dups', dup, 'inband-not-listed', (inband & ~lm).sum().item(),
          'listed gt', (lm & isgt).sum().item(), 'count>=definite rows bad', (c_gemm < definite.sum(1).int()).sum().item())
    ops.rank_resolve(st, S)
    ex = ((S64[:, a:b] > sg_all[:, None]) & ~isgt).sum(1).int()
    print('   final mismatch rows', (st.count != ex).sum().item(), (st.count - ex)[(st.count != ex)][:10].tolist())
