import sys, numpy as np, torch
sys.path.insert(0, '.')
from laff_amd import ops
sys.path.insert(0, 'tests')
from test_gpu_kernels import _exact_scores_f64, _count_ranks, rnd, dev
Nt, Nv, H, d = 1500, 1100, 2, 256
g = rnd(1000 + Nt + H)
zc = g.normal(0, 1, (17, H, d))
t = (zc[g.integers(0, 17, Nt)] + 0.15 * g.normal(0, 1, (Nt, H, d))).astype(np.float32)
v = (zc[g.integers(0, 17, Nv)] + 0.15 * g.normal(0, 1, (Nv, H, d))).astype(np.float32)
v[Nv // 2] = v[Nv // 3]
gt = dev(g.integers(0, Nv, Nt).astype(np.int32), torch.int32)
Et, Ev = dev(t), dev(v)
for prec in ('fp16', 'fp16x3'):
    T, V = ops.pack_rows(Et, True, 1e-13, prec), ops.pack_rows(Ev, True, 1e-13, prec)
    st = ops.rank_prepare(Et, Ev, T, V, gt)
    S = ops.sim_gemm_banded(st)
    c_gemm = st.count.clone()
    n, ov = st.listed_pairs()
    ops.rank_resolve(st, S)
    S64 = _exact_scores_f64(Et, Ev)
    want = _count_ranks(S64, gt)
    diff = (st.count + 1 - want)
    print(prec, 'listed', n, ov, 'diff hist', torch.unique(diff, return_counts=True))
    sg64 = S64.gather(1, gt.long()[:, None])[:, 0]
    print(' s_gt64 err', (st.s_gt64 - sg64).abs().max().item())
    plain = ops.sim_gemm(T, V, heads=H)
    band = st.band_t[:Nt, None] + st.band_v[None, :Nv]
    err = (plain.double() - S64).abs()
    print(' max err', err.max().item(), 'band min/max', band.min().item(), band.max().item(), 'ratio', (err / band).max().item())
    # definite counts the GEMM should have produced
    dl = plain - st.s_gt64.float()[:, None]
    definite = (dl > band)
    definite[torch.arange(Nt), gt.long()] = False
    print(' gemm definite count mismatch rows', (definite.sum(1).int() != c_gemm).sum().item(), (definite.sum(1).int() - c_gemm)[:10])
    inband = (dl.abs() <= band)
    inband[torch.arange(Nt), gt.long()] = False
    print(' inband pairs', inband.sum().item())
    pr = st.pair_indices()
    lm = torch.zeros_like(inband)
    lm[pr[:, 0], pr[:, 1]] = True
    print(' listed==inband', torch.equal(lm, inband), 'listed-not-inband', (lm & ~inband).sum().item(), 'inband-not-listed', (inband & ~lm).sum().item())
