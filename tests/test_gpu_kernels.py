"""Kernel-level parity on a real MI355X: every C-ABI entry point against the CPU oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from oracle import laff_oracle as O
from util import maxdiff

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def rnd(seed):
    return np.random.default_rng(seed)


# ---------------------------------------------------------------------------------------------- a1
@pytest.mark.parametrize('N,Dk,D', [(1, 4, 4), (33, 96, 512), (130, 1030, 260), (257, 2048, 384), (64, 512, 4096),
                                    (70, 77, 130), (5, 3981, 200)])
@pytest.mark.parametrize('act,bn', [('tanh', True), (None, False), ('relu', True), ('sigmoid', False)])
def test_fc_act_bn_vs_oracle(N, Dk, D, act, bn):
    from laff_amd import ops
    g = rnd(N * 7 + Dk)
    x = g.normal(0, 1, (N, Dk)).astype(np.float32)
    W = (g.normal(0, 1, (D, Dk)) / np.sqrt(Dk)).astype(np.float32)
    b = g.normal(0, 0.1, D).astype(np.float32)
    scale = g.uniform(0.5, 1.5, D).astype(np.float32) if bn else None
    shift = g.normal(0, 0.1, D).astype(np.float32) if bn else None
    y = ops.fc_act_bn(dev(x), dev(W), dev(b), dev(scale) if bn else None, dev(shift) if bn else None, act)
    ref = O.activation((x.astype(np.float64) @ W.astype(np.float64).T + b).astype(np.float32), act)
    if bn:
        ref = ref * scale + shift
    assert maxdiff(y, ref) <= 2e-5     # fp32 fma-chain vs fp64-accumulated reference, |y| <= ~6


def test_fc_act_bn_golden(golden):
    from laff_amd import ops
    g = golden('transform_net')
    for c in g.json('cases'):
        if not c['fc']:
            continue
        k = c['key']
        sd = g.sub(k + '/sd/')
        scale = shift = None
        if c['batch_norm']:
            scale = sd['bn1.weight'] / np.sqrt(sd['bn1.running_var'] + 1e-5)
            shift = sd['bn1.bias'] - sd['bn1.running_mean'] * scale
        y = ops.fc_act_bn(dev(g[k + '/x']), dev(sd['fc1.weight']), dev(sd['fc1.bias']),
                          dev(scale) if scale is not None else None, dev(shift) if shift is not None else None,
                          c['activation'])
        assert maxdiff(y, g[k + '/y']) <= 2e-5


def test_fc_strided_views():
    """ld != width on every operand (row-slices of wider buffers)."""
    from laff_amd import ops
    g = rnd(5)
    xb = dev(g.normal(0, 1, (40, 96)).astype(np.float32))
    wb = dev(g.normal(0, 1, (50, 80)).astype(np.float32))
    out = torch.zeros((40, 72), device=DEV)
    ops.fc_act_bn(xb[:, :64], wb[:, :64], out=out[:, 8:58])
    ref = xb[:, :64].double().cpu().numpy() @ wb[:, :64].double().cpu().numpy().T
    assert maxdiff(out[:, 8:58], ref.astype(np.float32)) <= 2e-5
    assert float(out[:, :8].abs().max()) == 0 and float(out[:, 58:].abs().max()) == 0


# ---------------------------------------------------------------------------------------------- a5/a6
def test_attention_1_golden(golden):
    from laff_amd import ops
    g = golden('attention_1')
    for c in g.json('cases'):
        k = c['key']
        x = dev(g[k + '/x'] if c.get('own_x') else g['x'])
        planes = [(x[:, l, :], False, None, None) for l in range(x.shape[1])]
        flags = ops.attention_flags(c['with_ave'], c['mul'])
        E, aw = ops.fuse(planes, 1, 512, dev(g[k + '/w']).view(1, 512), dev(g[k + '/b']).view(1),
                         dev(np.float32(c['gw'])).view(1), flags, return_weights=True)
        assert maxdiff(E[:, 0, :], g[k + '/out']) <= 2e-6
        if k + '/weights' in g and not c['with_ave']:
            assert maxdiff(aw[:, 0, :], g[k + '/weights']) <= 2e-6
    x = dev(g['x'])
    planes = [(x[:, l, :], False, None, None) for l in range(x.shape[1])]
    E = ops.fuse(planes, 1, 512, None, None, None, ops.attention_flags(just_average=True))
    assert maxdiff(E[:, 0, :], g['just_average/out']) <= 1e-6


def test_multi_head_golden(golden):
    from laff_amd import ops
    g = golden('multi_head')
    for c in g.json('cases'):
        k = c['key']
        sd = g.sub(k + '/sd/')
        att = O.attention_from_sd(sd, '', c['H'], c['with_ave'], c['mul'], c['split_head'], c['l2norm_each_head'])
        x = dev(g[k + '/x'])
        planes = [(x[:, l, :], False, None, None) for l in range(x.shape[1])]
        d = c['D'] // c['H'] if c['split_head'] else c['D']
        flags = ops.attention_flags(c['with_ave'], c['mul'], c['l2norm_each_head'], c['split_head'])
        E = ops.fuse(planes, c['H'], d, dev(att['w']), dev(att['b']), dev(att['gw']), flags)
        assert maxdiff(E, g[k + '/out']) <= 2e-6, c


@pytest.mark.parametrize('L', [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize('H,d', [(8, 512), (1, 512), (2, 256), (4, 64), (1, 1024), (2, 2048)])
def test_fuse_vs_oracle_with_tiled_planes(L, H, d):
    """Mixed FC planes and tiled no-transform planes with folded BN (model/model.py:1801-1805,1822-1823)."""
    from laff_amd import ops
    g = rnd(100 * L + H + d)
    N, D = 19, H * d
    planes_np, planes = [], []
    for l in range(L):
        if l % 2 == 1:
            x = g.normal(0, 1, (N, d)).astype(np.float32)
            sc = g.uniform(0.5, 1.5, D).astype(np.float32)
            sh = g.normal(0, 0.1, D).astype(np.float32)
            planes_np.append(np.tile(x, (1, H)) * sc + sh)
            planes.append((dev(x), True, dev(sc), dev(sh)))
        else:
            y = np.tanh(g.normal(0, 1, (N, D))).astype(np.float32)
            planes_np.append(y)
            planes.append((dev(y), False, None, None))
    w = g.uniform(-1, 1, (H, d)).astype(np.float32) / np.sqrt(d)
    b = g.normal(0, 0.3, H).astype(np.float32)
    gw = g.uniform(0, 1, H).astype(np.float32)
    for with_ave, mul in ((False, False), (True, True)):
        ref = O.multi_head_attention(np.stack(planes_np, 1), w, b, gw, H, with_ave, mul)
        E = ops.fuse(planes, H, d, dev(w), dev(b), dev(gw), ops.attention_flags(with_ave, mul))
        assert maxdiff(E, ref) <= 3e-6


# ---------------------------------------------------------------------------------------------- a7
@pytest.mark.parametrize('flags_name', ['attention_noAveNoAverageMul', 'average_AverageMul_noAve',
                                        'attention_noAverageMul_Ave', 'attention_averageMul'])
@pytest.mark.parametrize('d', [512, 64, 1024])
def test_frame_fuse_vs_oracle(flags_name, d):
    from laff_amd import ops
    with_ave, mul = O.FRAME_ATTENTION_FLAGS[flags_name]
    g = rnd(len(flags_name) + d)
    B, Fmax = 23, 17
    lens = g.integers(1, Fmax + 1, B).astype(np.int32)
    lens[0], lens[1] = Fmax, 1
    frames = np.zeros((B, Fmax, d), np.float32)
    for i in range(B):
        frames[i, :lens[i]] = g.normal(0, 1, (lens[i], d))
    w = (g.uniform(-1, 1, d) / np.sqrt(d)).astype(np.float32)
    b, gw = np.float32(0.37), np.float32(0.7)
    ref = O.frame_attention(frames, w, b, with_ave, mul, gw)
    flags = ops.attention_flags(with_ave, mul)
    out_masked = ops.frame_fuse(dev(frames), dev(lens, torch.int32), dev(w), dev(b).view(1), dev(gw).view(1), flags)
    out_full = ops.frame_fuse(dev(frames), None, dev(w), dev(b).view(1), dev(gw).view(1), flags)
    assert maxdiff(out_masked, ref) <= 2e-6
    assert maxdiff(out_full, ref) <= 2e-6


# ---------------------------------------------------------------------------------------------- a8-a11
PREC_TOL = {'fp32': 2e-6, 'fp16x3': 2e-6, 'bf16x3': 5e-6, 'fp16': 1e-4, 'bf16': 2e-3}


@pytest.mark.parametrize('precision', list(PREC_TOL))
@pytest.mark.parametrize('Nt,Nv,H,d', [(41, 29, 1, 512), (300, 257, 8, 512), (129, 128, 2, 64), (1, 1, 1, 64)])
def test_txt2vis_vs_oracle(precision, Nt, Nv, H, d):
    from laff_amd import ops
    g = rnd(Nt + Nv + H)
    t = g.normal(0, 1, (Nt, H, d)).astype(np.float32)
    v = g.normal(0, 2, (Nv, H, d)).astype(np.float32)
    ref = O.txt2vis_matrix(t, v)
    T = ops.pack_rows(dev(t), True, 1e-13, precision)
    V = ops.pack_rows(dev(v), True, 1e-13, precision)
    S = ops.sim_gemm(T, V, heads=H)
    tol = PREC_TOL[precision]
    if precision == 'fp16' and d < 512:
        tol = 4e-4     # the 1e-4 contract is stated at d = 512; fewer, larger components round coarser
    assert maxdiff(S, ref) <= tol


def test_txt2vis_golden(golden):
    from laff_amd import loss, ops
    g = golden('txt2vis')
    assert maxdiff(loss.cosine_sim(dev(g['t2']), dev(g['v2']), 'fp16x3'), g['s2']) <= 2e-6
    assert maxdiff(loss.cosine_sim(dev(g['t2']), dev(g['v2']), 'fp16'), g['s2']) <= 1e-4
    assert maxdiff(loss.l2norm(dev(g['l2/x'])), g['l2/default']) <= 1e-6
    for prec, tol in (('fp32', 2e-6), ('fp16', 1e-4)):
        T = ops.pack_rows(dev(g['t3']), True, 1e-13, prec)
        V = ops.pack_rows(dev(g['v3']), True, 1e-13, prec)
        assert maxdiff(ops.sim_gemm(T, V, heads=8), g['s3']) <= tol
    T = ops.pack_rows(dev(g['t3u']), True, 1e-13, 'fp32')
    V = ops.pack_rows(dev(g['v3u']), True, 1e-13, 'fp32')
    assert maxdiff(ops.sim_gemm(T, V, heads=4), g['s3u']) <= 2e-6


def test_sim_gemm_is_transpose_detecting():
    """Asymmetric operands: S[t, v] must pair text row t with video row v (guide rule 16)."""
    from laff_amd import ops
    Nt, Nv, K = 200, 136, 64
    t = np.zeros((Nt, K), np.float32)
    v = np.zeros((Nv, K), np.float32)
    t[:, 0] = np.arange(Nt) + 1
    v[:, 0] = 1.0 / (np.arange(Nv) + 1)
    T = ops.pack_rows(dev(t), False, 0.0, 'fp32', 1.0)
    V = ops.pack_rows(dev(v), False, 0.0, 'fp32', 1.0)
    S = ops.sim_gemm(T, V)
    ref = np.outer(t[:, 0], v[:, 0]).astype(np.float32)
    assert maxdiff(S, ref) <= 1e-4 * 200


# ---------------------------------------------------------------------------------------------- a12/a13
def test_rank_kernels_golden(golden):
    from laff_amd import predictor
    g = golden('eval')
    for c in g.json('cases'):
        k = c['key']
        t2v, v2t = predictor.retrieval_metrics(dev(g[k + '/S']), g.json(k + '/txt_ids'), g.json(k + '/vis_ids'))
        np.testing.assert_allclose(t2v, g[k + '/t2v'], rtol=0, atol=1e-12)
        np.testing.assert_allclose(v2t, g[k + '/v2t'], rtol=0, atol=1e-12)


def test_fused_rank_count_equals_unfused():
    from laff_amd import ops
    g = rnd(77)
    Nt, Nv, K = 515, 391, 128
    t = g.normal(0, 1, (Nt, K)).astype(np.float32)
    v = g.normal(0, 1, (Nv, K)).astype(np.float32)
    gt = (np.arange(Nt) % Nv).astype(np.int32)
    T = ops.pack_rows(dev(t), True, 1e-13, 'fp16')
    V = ops.pack_rows(dev(v), True, 1e-13, 'fp16')
    S = ops.sim_gemm(T, V)
    gtd = dev(gt, torch.int32)
    s_gt = ops.gather_gt(S, gtd)
    cnt_ref = ops.rank_count(S, gtd, s_gt)
    cnt = torch.zeros(Nt, dtype=torch.int32, device=DEV)
    S2 = ops.sim_gemm(T, V, gt_col=gtd, s_gt=s_gt, count=cnt)
    assert torch.equal(S, S2)
    assert torch.equal(cnt, cnt_ref)
    cnt2 = torch.zeros(Nt, dtype=torch.int32, device=DEV)
    ops.sim_gemm(T, V, want_scores=False, gt_col=gtd, s_gt=s_gt, count=cnt2)
    assert torch.equal(cnt2, cnt_ref)
    Sn = S.cpu().numpy()
    ref = np.array([np.sum(np.delete(Sn[i], gt[i]) > Sn[i, gt[i]]) for i in range(Nt)])
    assert np.array_equal(cnt_ref.cpu().numpy(), ref)


def test_sharded_rank_count_sums_to_global():
    """Column shards with col0 offsets + max/sum reductions reproduce the single-device ranks."""
    from laff_amd import ops
    g = rnd(78)
    Nt, Nv = 300, 260
    S = dev(g.normal(0, 1, (Nt, Nv)).astype(np.float32))
    gt = dev((np.arange(Nt) * 7 % Nv).astype(np.int32), torch.int32)
    full = ops.rank_count(S, gt, ops.gather_gt(S, gt))
    bounds = [0, 65, 130, 195, 260]
    parts = [S[:, a:b].contiguous() for a, b in zip(bounds[:-1], bounds[1:])]
    s_gt = torch.stack([ops.gather_gt(p, gt, col0=a) for p, a in zip(parts, bounds[:-1])]).max(dim=0).values
    total = torch.zeros(Nt, dtype=torch.int32, device=DEV)
    for p, a in zip(parts, bounds[:-1]):
        total += ops.rank_count(p, gt, s_gt, col0=a)
    assert torch.equal(total, full)


def test_fused_pipeline_is_self_consistent():
    """row_dot_gt pre-pass + fused count: the rank recounted from the written S equals the fused rank EXACTLY,
    and S[t, gt] carries the pre-pass value."""
    from laff_amd import ops
    g = rnd(79)
    Nt, Nv, K = 700, 333, 512
    t = g.normal(0, 1, (Nt, K)).astype(np.float32)
    v = g.normal(0, 1, (Nv, K)).astype(np.float32)
    gt = dev((np.arange(Nt) * 5 % Nv).astype(np.int32), torch.int32)
    for prec in ('fp16', 'bf16', 'fp16x3', 'bf16x3'):
        T = ops.pack_rows(dev(t), True, 1e-13, prec)
        V = ops.pack_rows(dev(v), True, 1e-13, prec)
        s_gt = ops.row_dot_gt(T, V, gt)
        plain = ops.sim_gemm(T, V)
        assert maxdiff(s_gt, ops.gather_gt(plain, gt)) <= 2e-6, prec       # same products, different fp32 order
        cnt = torch.zeros(Nt, dtype=torch.int32, device=DEV)
        S = ops.sim_gemm(T, V, gt_col=gt, s_gt=s_gt, count=cnt)
        assert torch.equal(ops.gather_gt(S, gt), s_gt)
        assert torch.equal(ops.rank_count(S, gt, s_gt), cnt)
        mask = torch.ones_like(S, dtype=torch.bool)
        mask[torch.arange(Nt, device=DEV), gt.long()] = False
        assert torch.equal(S[mask], plain[mask])
    # shard semantics: columns outside the shard give -inf
    T = ops.pack_rows(dev(t), True, 1e-13, 'fp16')
    V = ops.pack_rows(dev(v[100:200]), True, 1e-13, 'fp16')
    s = ops.row_dot_gt(T, V, gt, col0=100).cpu().numpy()
    inside = (gt.cpu().numpy() >= 100) & (gt.cpu().numpy() < 200)
    assert np.all(np.isneginf(s[~inside])) and np.all(np.isfinite(s[inside]))


def test_fc_grouped_equals_single():
    from laff_amd import ops
    g = rnd(80)
    probs, singles = [], []
    for (N, Dk, D, act) in [(300, 512, 512, 'tanh'), (77, 96, 260, None), (1, 4, 4, 'relu'), (513, 2048, 384, 'tanh'),
                            (40, 77, 130, 'sigmoid')]:
        x = dev(g.normal(0, 1, (N, Dk)).astype(np.float32))
        W = dev((g.normal(0, 1, (D, Dk)) / np.sqrt(Dk)).astype(np.float32))
        b = dev(g.normal(0, 0.1, D).astype(np.float32))
        sc = dev(g.uniform(0.5, 1.5, D).astype(np.float32))
        sh = dev(g.normal(0, 0.1, D).astype(np.float32))
        probs.append(dict(x=x, weight=W, bias=b, bn_scale=sc, bn_shift=sh, activation=act))
        singles.append(ops.fc_act_bn(x, W, b, sc, sh, act))
    outs = ops.fc_act_bn_grouped(probs)
    for a, b in zip(outs, singles):
        assert torch.equal(a, b)


@pytest.mark.parametrize('n,hi', [(1, 3000), (2, 3000), (3, 3000), (10, 3000), (1001, 3000), (40000, 3000), (40000, 2), (4096, 4097),
                                  (59800, 2990), (65536, 30000), (70001, 30000), (100000, 30000), (5000, 2 ** 31 - 1), (66000, 2 ** 24 + 5)])
def test_rank_metrics_device_vs_numpy(n, hi):
    from laff_amd import ops
    g = rnd(n + hi % 1000)
    r = g.integers(1, hi, n).astype(np.int32)
    if hi > 2:
        r[: n // 3] = 1
    got = ops.rank_metrics(dev(r, torch.int32))
    ranks = r.astype(np.float64)
    exp = (100.0 * np.mean(ranks <= 1), 100.0 * np.mean(ranks <= 5), 100.0 * np.mean(ranks <= 10), np.floor(np.median(ranks)),
           ranks.mean(), (1.0 / ranks).mean(), (1.0 / ranks).mean())
    np.testing.assert_allclose(got, exp, rtol=1e-13, atol=0)
    with pytest.raises(RuntimeError):
        ops.rank_metrics(dev(np.array([1, 0, 3], np.int32), torch.int32))


def _metrics_numpy(r):
    ranks = r.astype(np.float64)
    return (100.0 * np.mean(ranks <= 1), 100.0 * np.mean(ranks <= 5), 100.0 * np.mean(ranks <= 10), np.floor(np.median(ranks)),
            ranks.mean(), (1.0 / ranks).mean(), (1.0 / ranks).mean())


def test_rank_metrics_median_cases_of_the_histogram_paths():
    """The last block of the metrics launch takes the median out of the lo[] histogram (ranks < 256), out of hi[] + one pass, or out of
    the general select; even n needs sorted[k - 1] too, which may sit in another bin / byte / path.  A failed launch (rank < 1) must
    leave the scratch clean for the next one; n beyond 256 x 1024 makes every thread take several ranks."""
    from laff_amd import ops
    g = rnd(4242)
    cases = [np.array([3, 3, 7, 9], np.int32),                       # sorted[k-1] = 3, sorted[k] = 7: both in lo[]
             np.array([1, 1, 300, 400], np.int32),                   # below in lo[], median beyond it
             np.array([255, 256], np.int32), np.array([256, 255, 255, 256], np.int32),
             np.array([65279, 65280], np.int32), np.array([70000, 65279, 1, 65281], np.int32),
             np.array([511, 512, 512, 511], np.int32), np.array([5], np.int32), np.array([2 ** 31 - 1, 1], np.int32),
             g.integers(1, 200, 40000).astype(np.int32), g.integers(1, 200, 40001).astype(np.int32),
             g.integers(250, 262, 5000).astype(np.int32), g.integers(65270, 65290, 5000).astype(np.int32),
             g.integers(1, 50000, 300000).astype(np.int32), g.integers(1, 100, 1 << 19).astype(np.int32)]
    for r in cases:
        got = ops.rank_metrics(dev(r, torch.int32))
        np.testing.assert_allclose(got, _metrics_numpy(r), rtol=1e-13, atol=0, err_msg=str(r[:8]))
        with pytest.raises(RuntimeError):
            bad = r.copy(); bad[len(bad) // 2] = 0
            ops.rank_metrics(dev(bad, torch.int32))
        got2 = ops.rank_metrics(dev(r, torch.int32))
        assert tuple(got2) == tuple(got)                           # same bits: fixed reduction order, clean scratch


@pytest.mark.parametrize('N,Dk,D', [(1, 4, 4), (33, 96, 512), (130, 1030, 260), (300, 512, 512), (70, 77, 130), (5, 3981, 200)])
def test_fc_split_fp16x3_vs_fp64(N, Dk, D):
    """FC on the fp16 pipe with the exact hi/lo split: same tolerance as the fp32-MFMA path, incl. huge / tiny rows."""
    from laff_amd import ops
    g = rnd(N + Dk + D)
    x = g.normal(0, 1, (N, Dk)).astype(np.float32)
    x[0] *= 3e4            # beyond the fp16 range without the per-row scale
    if N > 2:
        x[1] *= 1e-6
        x[2] = 0
    W = (g.normal(0, 1, (D, Dk)) / np.sqrt(Dk)).astype(np.float32)
    b = g.normal(0, 0.1, D).astype(np.float32)
    sc = g.uniform(0.5, 1.5, D).astype(np.float32)
    sh = g.normal(0, 0.1, D).astype(np.float32)
    ws = ops.split_rows(dev(W))
    for act in (None, 'tanh'):
        y = ops.fc_act_bn_split_grouped([dict(x=dev(x), weight_split=ws, bias=dev(b), bn_scale=dev(sc), bn_shift=dev(sh),
                                              activation=act)])[0]
        pre = x.astype(np.float64) @ W.astype(np.float64).T + b
        ref = (np.tanh(pre) if act else pre) * sc + sh
        scale = 1.0 if act else np.maximum(1.0, np.abs(pre).max(axis=1, keepdims=True))
        assert float(np.max(np.abs(y.cpu().numpy() - ref) / scale)) <= 2e-5
        y32 = ops.fc_act_bn(dev(x), dev(W), dev(b), dev(sc), dev(sh), act)
        assert float(np.max(np.abs((y - y32).cpu().numpy()) / scale)) <= 2e-5


def test_fuse_emits_the_packed_operand():
    """laff_fuse_packed: the 16-bit similarity operand written by the fuse launch equals laff_pack_rows of its fp32
    output up to one fp16 ulp (E is unit-norm already, so the re-normalisation is a no-op at that precision)."""
    from laff_amd import ops
    g = rnd(81)
    N, H, d, L = 50, 8, 512, 4
    planes = [(dev(np.tanh(g.normal(0, 1, (N, H * d))).astype(np.float32)), False, None, None) for _ in range(L)]
    w = dev((g.uniform(-1, 1, (H, d)) / np.sqrt(d)).astype(np.float32))
    b = dev(g.normal(0, 0.3, H).astype(np.float32))
    for prec, dt in (('fp16', torch.float16), ('bf16', torch.bfloat16)):
        E, P = ops.fuse(planes, H, d, w, b, None, 0, packed_precision=prec)
        Q = ops.pack_rows(E, True, 1e-13, prec)
        a = P.buf[:N * H * d * 2].view(dt).float()
        c = Q.buf[:N * H * d * 2].view(dt).float()
        assert P.prescale == Q.prescale and P.K == Q.K == H * d
        ulp = (2.0 ** -10 if prec == 'fp16' else 2.0 ** -7)
        assert float(((a - c).abs() / (c.abs() + 1e-3)).max()) <= ulp
        S1 = ops.sim_gemm(P, P, heads=H)
        S2 = ops.sim_gemm(Q, Q, heads=H)
        assert maxdiff(S1, S2) <= (1e-4 if prec == 'fp16' else 2e-3)


@pytest.mark.parametrize('Nt,Nv,K', [(7, 30, 10), (50, 10000, 500), (9, 3000, 2000), (5, 33, 32), (3, 5, 1),
                                     (6, 80000, 500), (4, 50000, 3000), (3, 300000, 2000), (2, 20000, 8192)])   # wider than one LDS row: split + merge
def test_topk_rows_vs_argsort(Nt, Nv, K):
    from laff_amd import ops
    g = rnd(Nt + Nv + K)
    S = g.normal(0, 0.2, (Nt, Nv)).astype(np.float32)
    S[0, :min(5, Nv)] = S[0, min(7, Nv - 1)]     # ties, some of them across the K boundary for small K
    S[1 % Nt, -1] = np.float32(-0.0)
    if Nv > 40000:                               # ties between column blocks, at the top of the list
        S[Nt - 1, [11, 36000, Nv - 5]] = 9.0
    idx, val = ops.topk_rows(dev(S), K)
    ref = np.argsort(S, axis=1, kind='stable')[:, ::-1][:, :K]
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert np.array_equal(val.cpu().numpy(), np.take_along_axis(S, ref, axis=1))


def test_result_writers_golden(golden, tmp_path):
    """id.sent.score.txt and t2v.pkl byte-for-byte / value-for-value what the reference's writer produces."""
    import pickle
    from laff_amd import predictor
    g = golden('writers')
    S, vis_ids, txt_ids = dev(g['S']), g.json('vis_ids'), g.json('txt_ids')

    class _DS:
        def get_caption_dict_by_id(self, cid):
            return {'caption': 'caption of ' + cid}

    class _TL:
        dataset = _DS()

    for name, thr in (('top10', 10), ('all', 2000)):
        f, pk = str(tmp_path / (name + '.txt')), str(tmp_path / (name + '.pkl'))
        predictor.txt2video_write_to_file(f, S, vis_ids, txt_ids, pkl_saved_file=pk, txt_loader=_TL(), Threshold=thr)
        assert open(f).read() == str(g[name + '/text'])
        got = pickle.load(open(pk, 'rb'))
        exp = g.json(name + '/pkl')
        assert list(got.keys()) == list(exp.keys())
        for k in exp:
            assert got[k]['query'] == exp[k]['query'] and got[k]['rank_list'] == exp[k]['rank_list']
            assert [repr(float(x)) for x in got[k]['sim_value']] == exp[k]['sim_value']


def test_result_writers_from_operands_golden(golden, tmp_path):
    """The same two files WITHOUT a score matrix: the writer is handed packed GEMM operands (here T = the fixture's score rows and
    V = the identity, fp32: T.V^T reproduces the fixture's scores exactly) and takes the lists from ops.topk_from_operands, blocks
    of 7 texts at a time."""
    from laff_amd import ops, predictor
    g = golden('writers')
    S, vis_ids, txt_ids = g['S'], g.json('vis_ids'), g.json('txt_ids')
    Nt, Nv = S.shape
    Kp = (Nv + 15) & ~15
    t = np.zeros((Nt, Kp), np.float32); t[:, :Nv] = S
    v = np.zeros((Nv, Kp), np.float32); v[np.arange(Nv), np.arange(Nv)] = 1.0
    T, V = ops.pack_rows(dev(t), False, 1e-13, 'fp32'), ops.pack_rows(dev(v), False, 1e-13, 'fp32')
    assert torch.equal(ops.sim_gemm(T, V), dev(S))
    for name, thr in (('top10', 10), ('all', 2000)):
        f = str(tmp_path / (name + '.txt'))
        predictor.txt2video_write_to_file(f, (T, V, 1), vis_ids, txt_ids, Threshold=thr, block_rows=7)
        assert open(f).read() == str(g[name + '/text'])


@pytest.mark.parametrize('prec', ['fp16', 'fp16x3'])
def test_topk_from_operands_equals_topk_of_the_score_matrix(prec):
    """Row-blocked top-K straight from the operands == top-K of the materialised matrix, indices and scores, with exact ties
    (duplicated videos) inside and across the kept part."""
    from laff_amd import ops
    g = rnd(5)
    Nt, Nv, K = 1500, 2600, 512
    v = g.normal(0, 1, (Nv, K)).astype(np.float32)
    v[1300:1500] = v[100:300]                          # 200 exact duplicates: equal scores, larger index first
    t = (v[g.integers(0, Nv, Nt)] + 0.7 * g.normal(0, 1, (Nt, K))).astype(np.float32)
    T, V = ops.pack_rows(dev(t), True, 1e-13, prec), ops.pack_rows(dev(v), True, 1e-13, prec)
    S = ops.sim_gemm(T, V)
    for k in (1, 37, 700):
        i0, v0 = ops.topk_rows(S, k)
        i1, v1 = ops.topk_from_operands(T, V, k, block_rows=400)
        assert torch.equal(i0, i1) and torch.equal(v0, v1)
    i2, v2 = ops.topk_from_operands(T, V, 37)           # default block: one pass
    i0, v0 = ops.topk_rows(S, 37)
    assert torch.equal(i0, i2) and torch.equal(v0, v2)


# ---------------------------------------------------------------------------------------------- a1, sparse input (bow)
def _random_csr(g, N, Dk, max_nnz):
    rows = []
    for _ in range(N):
        k = int(g.integers(0, max_nnz + 1))                       # empty captions included
        ids = np.sort(g.choice(Dk, size=min(k, Dk), replace=False))
        rows.append((ids, g.integers(1, 4, size=ids.size).astype(np.float32)))
    crow = np.zeros(N + 1, np.int32)
    crow[1:] = np.cumsum([len(r[0]) for r in rows])
    col = np.concatenate([r[0] for r in rows]).astype(np.int32) if N else np.zeros(0, np.int32)
    val = np.concatenate([r[1] for r in rows]).astype(np.float32) if N else np.zeros(0, np.float32)
    dense = np.zeros((N, Dk), np.float32)
    for i, (ids, v) in enumerate(rows):
        dense[i, ids] = v
    return crow, col, val, dense


@pytest.mark.parametrize('N,Dk,D', [(1, 4, 4), (33, 333, 512), (130, 7811, 4096), (64, 1000, 2048), (17, 50, 260), (9, 77, 8192)])
@pytest.mark.parametrize('act,bn', [('tanh', True), (None, False), ('relu', True), ('sigmoid', False)])
def test_fc_gather_vs_oracle_on_densified_input(N, Dk, D, act, bn):
    from laff_amd import ops
    g = rnd(N * 5 + D)
    crow, col, val, dense = _random_csr(g, N, Dk, 24)
    W = (g.normal(0, 1, (D, Dk)) / 4).astype(np.float32)
    b = g.normal(0, 0.1, D).astype(np.float32)
    scale = g.uniform(0.5, 1.5, D).astype(np.float32) if bn else None
    shift = g.normal(0, 0.1, D).astype(np.float32) if bn else None
    x = torch.sparse_csr_tensor(dev(crow, torch.int32), dev(col, torch.int32), dev(val), size=(N, Dk))
    y = ops.fc_gather_act_bn(x, dev(W.T), dev(b), dev(scale) if bn else None, dev(shift) if bn else None, act)
    ref = O.transform_net(dense, W, b, act, None)                 # the reference densifies bow and runs nn.Linear
    if bn:
        ref = ref * scale + shift
    assert maxdiff(y, ref) <= 1e-5
    # and it agrees with the dense FC kernel on the same (densified) input
    yd = ops.fc_act_bn(dev(dense), dev(W), dev(b), dev(scale) if bn else None, dev(shift) if bn else None, act)
    assert maxdiff(y, yd) <= 1e-5


def test_fc_gather_ignores_out_of_vocabulary_ids_and_rejects_bad_shapes():
    from laff_amd import ops
    g = rnd(3)
    Dk, D = 40, 64
    W = g.normal(0, 1, (D, Dk)).astype(np.float32)
    crow = np.array([0, 3, 3, 5], np.int32)
    col = np.array([1, 39, 7, 0, 2], np.int32)
    val = np.ones(5, np.float32)
    x = torch.sparse_csr_tensor(dev(crow, torch.int32), dev(col, torch.int32), dev(val), size=(3, Dk))
    y = ops.fc_gather_act_bn(x, dev(W.T))
    ref = np.stack([W[:, [1, 39, 7]].sum(1), np.zeros(D, np.float32), W[:, [0, 2]].sum(1)])
    assert maxdiff(y, ref) <= 1e-5
    with pytest.raises(ValueError):
        ops.fc_gather_act_bn(x, dev(W.T[:30]))
    with pytest.raises(RuntimeError):
        ops.fc_gather_act_bn(dev(np.zeros((3, Dk), np.float32)), dev(W.T))


# ---------------------------------------------------------------------------------------------- 8f-4 training loss
def test_margin_loss_golden_forward_and_backward(golden):
    """laff_margin_loss against loss.MarginRankingLoss + autograd of the reference (per head, summed)."""
    from laff_amd import ops
    g = golden('margin_loss')
    for c in g.json('cases'):
        k = c['key']
        loss, d_s, d_im = ops.margin_loss(dev(g[k + '/s']), dev(g[k + '/im']), c['margin'], c['max_violation'], c['cost_style'],
                                          c['direction'])
        ref = float(g[k + '/loss'])
        assert abs(loss.item() - ref) <= 2e-5 * max(1.0, abs(ref)), c
        assert maxdiff(d_s, g[k + '/d_s']) <= 2e-6, c
        assert maxdiff(d_im, g[k + '/d_im']) <= 2e-6, c


@pytest.mark.parametrize('B,H,d', [(128, 8, 512), (256, 1, 512), (1, 1, 8), (2, 3, 5), (130, 2, 30), (512, 4, 64)])
@pytest.mark.parametrize('maxv,style,direction', [(True, 'sum', 't2i'), (False, 'sum', 'bidir'), (True, 'mean', 'bidir'), (False, 'mean', 'i2t')])
def test_margin_loss_vs_oracle(B, H, d, maxv, style, direction):
    from laff_amd import ops
    g = rnd(B + H + d)
    z = g.normal(0, 1, (B, 16)).astype(np.float32)
    P = g.normal(0, 1, (16, H * d)).astype(np.float32)
    s = (z @ P + 2.0 * g.normal(0, 1, (B, H * d))).astype(np.float32).reshape(B, H, d)
    im = (z @ P + 2.0 * g.normal(0, 1, (B, H * d))).astype(np.float32).reshape(B, H, d)
    loss, d_s, d_im = ops.margin_loss(dev(s), dev(im), 0.2, maxv, style, direction)
    rl, rs, ri = O.margin_ranking_loss(s, im, 0.2, maxv, style, direction)
    assert abs(loss.item() - float(rl)) <= 5e-5 * max(1.0, abs(float(rl)))
    # a hinge / arg-max decision that flips on a 1-ulp score difference moves whole rows of the gradient: compare robustly
    bad_s = (np.abs(d_s.cpu().numpy() - rs).max(axis=-1) > 5e-6).mean()
    bad_i = (np.abs(d_im.cpu().numpy() - ri).max(axis=-1) > 5e-6).mean()
    assert bad_s <= 0.01 and bad_i <= 0.01, (bad_s, bad_i)
    # forward-only call leaves no gradient buffers
    l2, a, b = ops.margin_loss(dev(s), dev(im), 0.2, maxv, style, direction, want_grad=False)
    assert a is None and b is None and abs(l2.item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))


def test_margin_loss_module_plugs_into_autograd_and_2d_inputs():
    from laff_amd import loss as L
    g = rnd(77)
    s = torch.tensor(g.normal(0, 1, (40, 64)).astype(np.float32), device=DEV, requires_grad=True)
    im = torch.tensor(g.normal(0, 1, (40, 64)).astype(np.float32), device=DEV, requires_grad=True)
    crit = L.MarginRankingLoss(margin=0.2, max_violation=True, cost_style='sum', direction='t2i')
    loss, items = L.compute_loss(crit, im, s)
    (3.0 * loss).backward()
    rl, rs, ri = O.margin_ranking_loss(s.detach().cpu().numpy(), im.detach().cpu().numpy(), 0.2, True, 'sum', 't2i')
    assert abs(loss.item() - float(rl)) <= 2e-5 * max(1.0, float(rl)) and items['triplet_loss'] is loss
    assert maxdiff(s.grad, 3.0 * rs) <= 1e-5 and maxdiff(im.grad, 3.0 * ri) <= 1e-5
    with pytest.raises(NotImplementedError):
        L.MarginRankingLoss(measure='hist')
    with pytest.raises(ValueError):
        from laff_amd import ops
        ops.margin_loss(s.detach(), im.detach()[:, :32], 0.2)


# ---------------------------------------------------------------------------------------------- empty / degenerate inputs
def test_empty_inputs_are_accepted_everywhere():
    """N = 0 on any side is legal in the reference (an empty last batch / an empty query set): nothing is launched, shapes hold."""
    from laff_amd import ops
    W = dev(rnd(1).normal(0, 1, (32, 16)).astype(np.float32))
    y = ops.fc_act_bn(torch.empty((0, 16), device=DEV), W, None, None, None, 'tanh')
    assert y.shape == (0, 32)
    E = ops.fuse([(torch.empty((0, 64), device=DEV), False, None, None)] * 2, 1, 64, dev(np.zeros((1, 64))), dev(np.zeros(1)),
                 dev(np.zeros(1)), ops.attention_flags())
    assert E.shape[0] == 0
    T = ops.pack_rows(torch.empty((0, 64), device=DEV), True, 1e-13, 'fp16')
    V = ops.pack_rows(dev(rnd(2).normal(0, 1, (5, 64)).astype(np.float32)), True, 1e-13, 'fp16')
    assert ops.sim_gemm(T, V).shape == (0, 5)
    assert ops.sim_gemm(V, T).shape == (5, 0)
    S = torch.empty((0, 5), device=DEV)
    gt = torch.empty((0,), dtype=torch.int32, device=DEV)
    assert ops.rank_count(S, gt, ops.gather_gt(S, gt)).shape == (0,)
    x = torch.sparse_csr_tensor(torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(0, dtype=torch.int32, device=DEV),
                                torch.zeros(0, device=DEV), size=(0, 16))
    assert ops.fc_gather_act_bn(x, dev(np.zeros((16, 32)))).shape == (0, 32)


def test_zero_vectors_follow_the_reference_l2norm():
    """An all-zero embedding has norm 0: l2norm divides by eps + 1e-14 and yields zeros, so its scores are exactly 0
    (loss.py:8-13) -- on every precision path."""
    from laff_amd import ops
    g = rnd(5)
    t = g.normal(0, 1, (9, 64)).astype(np.float32)
    v = g.normal(0, 1, (7, 64)).astype(np.float32)
    t[3] = 0
    v[0] = 0
    ref = O.txt2vis_matrix(t, v)
    assert np.all(ref[3] == 0) and np.all(ref[:, 0] == 0)
    for prec, tol in PREC_TOL.items():
        S = ops.sim_gemm(ops.pack_rows(dev(t), True, 1e-13, prec), ops.pack_rows(dev(v), True, 1e-13, prec)).cpu().numpy()
        assert np.all(S[3] == 0) and np.all(S[:, 0] == 0), prec
        assert np.abs(S - ref).max() <= max(tol, 4e-4), prec


def test_frame_fuse_degenerate_videos():
    """A video whose frames are all padding (len 0) and a one-frame video, against the oracle (model/model.py:2147-2190)."""
    from laff_amd import ops
    g = rnd(8)
    B, F, d = 5, 6, 64
    lens = np.array([0, 1, 6, 3, 0], np.int32)
    frames = np.zeros((B, F, d), np.float32)
    for i in range(B):
        frames[i, :lens[i]] = g.normal(0, 1, (lens[i], d))
    w = g.normal(0, 0.2, d).astype(np.float32)
    b, gw = np.float32(0.1), np.float32(0.6)
    for name, (with_ave, mul) in O.FRAME_ATTENTION_FLAGS.items():
        got = ops.frame_fuse(dev(frames), dev(lens, torch.int32), dev(w), dev(b).view(1), dev(gw).view(1),
                             ops.attention_flags(with_ave, mul)).cpu().numpy()
        ref = O.frame_attention(frames, w, b, with_ave, mul, gw)
        assert np.array_equal(np.isfinite(got), np.isfinite(ref)), name
        m = np.isfinite(ref)
        assert np.abs(got[m] - ref[m]).max() <= 2e-6, name


@pytest.mark.parametrize('act', ['tanh', 'relu', 'sigmoid'])
def test_fuse_plane_activation_equals_activation_in_the_projection(act):
    """plane.act: a projection may hand over x W^T + b and leave activation + BatchNorm to laff_fuse -- same bits."""
    from laff_amd import ops
    g = rnd(31)
    N, Dk, H, d = 70, 96, 2, 64
    D = H * d
    xs = [dev(g.normal(0, 1, (N, Dk)).astype(np.float32)) for _ in range(3)]
    Ws = [dev((g.normal(0, 1, (D, Dk)) / 8).astype(np.float32)) for _ in range(3)]
    bs = [dev(g.normal(0, 0.1, D).astype(np.float32)) for _ in range(3)]
    sc = [dev(g.uniform(0.5, 1.5, D).astype(np.float32)) for _ in range(3)]
    sh = [dev(g.normal(0, 0.1, D).astype(np.float32)) for _ in range(3)]
    w, b, gw = dev(g.normal(0, 0.2, (H, d)).astype(np.float32)), dev(g.normal(0, 0.1, H).astype(np.float32)), dev(np.full(H, 0.6, np.float32))
    flags = ops.attention_flags(with_ave=True)
    fused_in_fc = [(ops.fc_act_bn(x, W, bb, s, t, act), False, None, None) for x, W, bb, s, t in zip(xs, Ws, bs, sc, sh)]
    deferred = [(ops.fc_act_bn(x, W, bb, None, None, None), False, s, t, act) for x, W, bb, s, t in zip(xs, Ws, bs, sc, sh)]
    E1 = ops.fuse(fused_in_fc, H, d, w, b, gw, flags)
    E2 = ops.fuse(deferred, H, d, w, b, gw, flags)
    assert torch.equal(E1, E2)


def test_frame_fuse_grouped_equals_per_feature_launches():
    from laff_amd import ops
    g = rnd(12)
    B, Fmax, d = 37, 9, 128
    lens = g.integers(0, Fmax + 1, B).astype(np.int32)
    frames, params = [], []
    for _ in range(5):
        f = np.zeros((B, Fmax, d), np.float32)
        for i in range(B):
            f[i, :lens[i]] = g.normal(0, 1, (lens[i], d))
        frames.append(dev(f))
        params.append((dev(g.normal(0, 0.2, d).astype(np.float32)), dev(g.normal(0, 0.2, 1).astype(np.float32)),
                       dev(g.uniform(0, 1, 1).astype(np.float32))))
    ld = dev(lens, torch.int32)
    for with_ave, mul in O.FRAME_ATTENTION_FLAGS.values():
        flags = ops.attention_flags(with_ave, mul)
        grouped = ops.frame_fuse_grouped(frames, ld, params, flags)
        for f, (w, b, gw), v in zip(frames, params, grouped):
            assert torch.equal(v, ops.frame_fuse(f, ld, w, b, gw, flags))
        # the reference's mask_tensor instead of lens (summed inside the launch), also as a view into a wider buffer, Fmax beyond a wave
        mask = (torch.arange(Fmax, device='cuda')[None, :] < ld[:, None]).to(torch.float32)
        wide = torch.zeros((B, Fmax + 7), device='cuda')
        wide[:, :Fmax] = mask
        for m in (mask, wide[:, :Fmax]):
            for v, w in zip(grouped, ops.frame_fuse_grouped(frames, None, params, flags, mask=m)):
                assert torch.equal(v, w)
    B2, F2, d2 = 5, 150, 64
    l2 = np.array([0, 1, 64, 65, 150], np.int32)
    f2 = np.zeros((B2, F2, d2), np.float32)
    for i in range(B2):
        f2[i, :l2[i]] = g.normal(0, 1, (l2[i], d2))
    fr2, pr2 = [dev(f2)], [params[0][:0] + (dev(g.normal(0, 0.2, d2).astype(np.float32)), params[0][1], params[0][2])]
    flags = ops.attention_flags(True, False)
    a = ops.frame_fuse_grouped(fr2, dev(l2, torch.int32), pr2, flags)[0]
    m2 = (torch.arange(F2, device='cuda')[None, :] < dev(l2, torch.int32)[:, None]).to(torch.float32)
    assert torch.equal(a, ops.frame_fuse_grouped(fr2, None, pr2, flags, mask=m2)[0])
    with pytest.raises(ValueError):
        ops.frame_fuse_grouped(fr2, None, pr2, flags, mask=m2[:, :F2 - 1])                    # a mask narrower than the frames
    with pytest.raises(ValueError):
        ops.frame_fuse_grouped(fr2, None, pr2, flags, mask=m2.t().contiguous().t())          # no unit column stride


@pytest.mark.parametrize('H,d,act', [(8, 512, 'tanh'), (1, 64, None), (2, 256, 'relu'), (4, 128, 'sigmoid')])
def test_fuse_gather_plane_equals_separate_gather_fc(H, d, act):
    """A sparse feature through its FC as a GATHER plane of laff_fuse == laff_fc_gather_act_bn followed by a plain plane."""
    from laff_amd import ops
    g = rnd(H * d)
    N, Dk, D = 75, 300, H * d
    crow, col, val, dense = _random_csr(g, N, Dk, 20)
    W = (g.normal(0, 1, (D, Dk)) / 4).astype(np.float32)
    bias = g.normal(0, 0.1, D).astype(np.float32)
    sc = g.uniform(0.5, 1.5, D).astype(np.float32)
    sh = g.normal(0, 0.1, D).astype(np.float32)
    other = np.tanh(g.normal(0, 1, (N, D))).astype(np.float32)
    clip = g.normal(0, 1, (N, d)).astype(np.float32)
    w, b, gw = dev(g.normal(0, 0.2, (H, d)).astype(np.float32)), dev(g.normal(0, 0.1, H).astype(np.float32)), dev(np.full(H, 0.6, np.float32))
    csr = torch.sparse_csr_tensor(dev(crow, torch.int32), dev(col, torch.int32), dev(val), size=(N, Dk))
    wt = dev(np.ascontiguousarray(W.T))
    flags = ops.attention_flags(with_ave=True)
    y = ops.fc_gather_act_bn(csr, wt, dev(bias), dev(sc), dev(sh), act)
    base = [(dev(other), False, None, None), (dev(clip), True, dev(sc), dev(sh))]
    E_ref = ops.fuse([(y, False, None, None)] + base, H, d, w, b, gw, flags)
    E_gat = ops.fuse([(None, False, dev(sc), dev(sh), act, (csr, wt, dev(bias)))] + base, H, d, w, b, gw, flags)
    assert maxdiff(E_gat, E_ref) <= 2e-6
    # against the oracle on the densified input
    ref_plane = O.transform_net(dense, W, bias, act, None) * sc + sh
    ref = O.multi_head_attention(np.stack([ref_plane, other, np.tile(clip, (1, H)) * sc + sh], axis=1), w.cpu().numpy(), b.cpu().numpy(),
                                 gw.cpu().numpy(), H, True, False)
    assert maxdiff(E_gat, ref) <= 5e-6


# ---- exact ranks on a reduced-precision GEMM (laff_rank_prepare -> laff_sim_gemm_banded -> laff_rank_resolve) --------------
def _exact_scores_f64(t, v):
    """torch float64 statement of oracle.txt2vis_matrix_f64 on the device (the oracle itself checks it below at small sizes)."""
    t, v = t.double(), v.double()
    if t.dim() == 2:
        t, v = t[:, None, :], v[:, None, :]
    H = t.shape[1]
    S = torch.zeros((t.shape[0], v.shape[0]), dtype=torch.float64, device=t.device)
    for h in range(H):
        tn = t[:, h] / (t[:, h].pow(2).sum(1, keepdim=True).sqrt() + (1e-13 + 1e-14))
        vn = v[:, h] / (v[:, h].pow(2).sum(1, keepdim=True).sqrt() + (1e-13 + 1e-14))
        S += tn @ vn.T
    return S / H


def _count_ranks(S, gt):
    sg = S.gather(1, gt.long()[:, None])
    above = S > sg
    above[torch.arange(S.shape[0], device=S.device), gt.long()] = False
    return above.sum(1).to(torch.int32) + 1


@pytest.mark.parametrize('precision', ['fp16', 'bf16', 'fp16x3', 'bf16x3', 'fp32'])
@pytest.mark.parametrize('Nt,Nv,H,d', [(700, 333, 1, 512), (513, 257, 8, 64), (1500, 1100, 2, 256), (64, 9, 1, 36)])
def test_exact_ranks_equal_fp64_ranks(precision, Nt, Nv, H, d):
    """Whatever the operand precision, count + 1 are the ranks of the exact (fp64) cosine scores of the fp32 embeddings, the band
    really bounds the GEMM's error, the listed pairs are few, and the S that comes back recounts to the same ranks."""
    from laff_amd import ops
    from oracle import laff_oracle as O
    g = rnd(1000 + Nt + H)
    # clustered rows: many near ties around the ground-truth score (what makes a 16-bit GEMM mis-rank)
    zc = g.normal(0, 1, (17, H, d))
    t = (zc[g.integers(0, 17, Nt)] + 0.15 * g.normal(0, 1, (Nt, H, d))).astype(np.float32)
    v = (zc[g.integers(0, 17, Nv)] + 0.15 * g.normal(0, 1, (Nv, H, d))).astype(np.float32)
    v[Nv // 2] = v[Nv // 3]                                    # an exact duplicate video: a true tie, never counted
    gt = dev(g.integers(0, Nv, Nt).astype(np.int32), torch.int32)
    Et, Ev = dev(t), dev(v)
    T, V = ops.pack_rows(Et, True, 1e-13, precision), ops.pack_rows(Ev, True, 1e-13, precision)
    S, count, st = ops.exact_ranks(Et, Ev, T, V, gt)
    S64 = _exact_scores_f64(Et, Ev)
    # a float64 GEMM is not bit-identical across columns, so the duplicate video ties with its twin only to ~1e-16 in S64: rows
    # whose ground truth is one of the twins are compared with the twin excluded (the kernel scores equal rows identically)
    twins = torch.tensor([Nv // 2, Nv // 3], device=DEV)
    S64t = S64.clone()
    for a, b in ((0, 1), (1, 0)):
        rows = (gt.long() == twins[a]).nonzero()[:, 0]
        S64t[rows, twins[b]] = S64t[rows, twins[a]]
    want = _count_ranks(S64t, gt)
    assert torch.equal(count + 1, want), precision
    if Nt * Nv <= 200000:                                      # the device fp64 statement against the numpy oracle
        o64 = O.txt2vis_matrix_f64(t, v)
        gn = gt.cpu().numpy()
        for a, b in ((Nv // 2, Nv // 3), (Nv // 3, Nv // 2)):
            o64[gn == a, b] = o64[gn == a, a]
        assert np.array_equal(want.cpu().numpy(), O.count_ranks(o64, gn))
    n_listed, overflow = st.listed_pairs()
    assert not overflow
    # the band is a true bound on the plain GEMM's error, and not a loose one
    plain = ops.sim_gemm(T, V, heads=H)
    band = st.band_t[:Nt, None] + st.band_v[None, :Nv]
    ratio = ((plain.double() - S64).abs() / band.double()).max().item()
    assert ratio <= 1.0, (precision, ratio)
    assert n_listed <= int((((plain.double() - S64.gather(1, gt.long()[:, None])).abs() <= 2 * band.double()).sum().item())) + Nt
    # S: exact value at the ground truth, rank-consistent everywhere
    assert torch.equal(ops.gather_gt(S, gt), st.s_gt64.float())
    assert torch.equal(ops.rank_count(S, gt, ops.gather_gt(S, gt)), count)
    tol = {'fp16': 1e-4 * (1 + 22 / np.sqrt(d)), 'bf16': 8e-4 * (1 + 22 / np.sqrt(d)), 'fp16x3': 2e-6, 'bf16x3': 5e-6, 'fp32': 2e-6}[precision]
    assert (S.double() - S64).abs().max().item() <= tol


def test_exact_ranks_shard_semantics_and_overflow_flag():
    """Video shards: s_gt64 is -inf for texts whose video lives elsewhere; MAX over shards + SUM of counts = the global ranks.
    A pair list that is too small is reported, not silently truncated."""
    from laff_amd import ops
    g = rnd(77)
    Nt, Nv, H, d = 900, 400, 1, 128
    t = g.normal(0, 1, (Nt, H, d)).astype(np.float32)
    v = (g.normal(0, 1, (Nv, H, d)) * 0.05 + g.normal(0, 1, (1, H, d))).astype(np.float32)      # near-identical videos: many ties
    gt = dev(g.integers(0, Nv, Nt).astype(np.int32), torch.int32)
    Et, Ev = dev(t), dev(v)
    T = ops.pack_rows(Et, True, 1e-13, 'fp16')
    want = _count_ranks(_exact_scores_f64(Et, Ev), gt)
    bounds = [0, 130, 131, 400]
    states = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        Evs = Ev[a:b].contiguous()
        states.append(ops.rank_prepare(Et, Evs, T, ops.pack_rows(Evs, True, 1e-13, 'fp16'), gt, col0=a))
    s_all = torch.stack([s.s_gt64 for s in states]).max(dim=0).values
    gtc = gt.cpu().numpy()
    for s, a, b in zip(states, bounds[:-1], bounds[1:]):
        own = torch.from_numpy((gtc >= a) & (gtc < b)).to(DEV)
        assert torch.isneginf(s.s_gt64[~own]).all() and torch.isfinite(s.s_gt64[own]).all()
    total = torch.zeros(Nt, dtype=torch.int32, device=DEV)
    for s in states:
        s.s_gt64.copy_(s_all)
        Sb = ops.sim_gemm_banded(s)
        total += ops.rank_resolve(s, Sb)
    assert torch.equal(total + 1, want)
    # overflow: a list of 8 pairs cannot hold this problem's near ties
    st = ops.rank_prepare(Et, Ev, T, ops.pack_rows(Ev, True, 1e-13, 'fp16'), gt, pair_cap=8)
    ops.rank_resolve(st, ops.sim_gemm_banded(st))
    n, overflow = st.listed_pairs()
    assert 0 < n <= 8 and overflow          # (n: the pairs that fitted)
    with pytest.raises(RuntimeError, match='rank < 1'):
        ops.rank_metrics(st.count, base=1)


@pytest.mark.parametrize('strip', [False, True])
def test_overflow_is_flagged_through_the_fused_tail(strip):
    """An overflowing pair list driven through laff_rank_resolve_metrics (the one-launch tail): block 0's poison must not be needed by
    the finishing workgroup (it may sit in another XCD's L2) -- the tail derives the overflow from the list header itself.  Both list
    formats (the tiled kernel's pairs, the strip kernel's dumps), eager (RuntimeError) and replayed from a captured graph (flag)."""
    from laff_amd import ops
    g = rnd(78)
    if strip:
        Nt, Nv, H, d = 12288, 4096, 1, 512          # 48 strips x 128 column blocks = 24 units per CU: the strip kernel's threshold
    else:
        Nt, Nv, H, d = 900, 400, 1, 128
    t = g.normal(0, 1, (Nt, H, d)).astype(np.float32)
    v = (g.normal(0, 1, (Nv, H, d)) * 0.02 + g.normal(0, 1, (1, H, d))).astype(np.float32)      # near-identical videos: many ties
    gt = dev(g.integers(0, Nv, Nt).astype(np.int32), torch.int32)
    Et, Ev = dev(t), dev(v)
    T = ops.pack_rows(Et, True, 1e-13, 'fp16')
    V = ops.pack_rows(Ev, True, 1e-13, 'fp16')
    ops.ctx_prepare_metrics(torch.device(DEV))
    st = ops.rank_prepare(Et, Ev, T, V, gt, pair_cap=8)
    ops.sim_gemm_banded(st, want_scores=False)
    with pytest.raises(RuntimeError):
        ops.rank_resolve_metrics(st)
    assert st.listed_pairs()[1]
    pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode='thread_local'):
        st2 = ops.rank_prepare(Et, Ev, T, V, gt, pair_cap=8)
        ops.sim_gemm_banded(st2, want_scores=False)
        ops.rank_resolve_metrics(st2, None, pinned)
    for _ in range(5):
        pinned.fill_(-1.0)
        gr.replay()
        torch.cuda.synchronize()
        assert pinned[7].item() == 1.0 and np.isnan(pinned[0].item())


@pytest.mark.parametrize('mode', ['fused', 'split'])
def test_fence_free_tail_equals_the_fenced_build(mode, tmp_path):
    """The metrics tails hand partial results between workgroups without release / acquire fences (gfx9 hardware behaviour, see
    LAFF_TAIL_FENCES in rank.hip).  The memory model's own form stays one compile switch away: this test builds rank.hip with
    -DLAFF_TAIL_FENCES, and holds 150 graph replays of {prepare, banded GEMM, tail} on a large grid (1,536 resolve workgroups; 40 metrics
    workgroups for 'split') of the shipped library against the fenced build: ONE metrics tuple in each, equal to each other."""
    import json
    import os
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc on this box')
    libdir = os.path.join(root, 'laff_amd', 'lib')
    vdir = str(tmp_path)
    objs = []
    for o in os.listdir(libdir):
        if o.endswith('.o') and o != 'rank.o':
            objs.append(os.path.join(libdir, o))
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++20', '-fPIC', '-fno-gpu-rdc']
    r = subprocess.run([hipcc] + flags + ['-DLAFF_TAIL_FENCES', '-c', os.path.join(root, 'laff_amd', 'csrc', 'rank.hip'), '-o',
                                          os.path.join(vdir, 'rank.o')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    vlib = os.path.join(vdir, 'liblaff_hip.so')
    r = subprocess.run([hipcc, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', vlib, os.path.join(vdir, 'rank.o')] + objs,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    script = os.path.join(root, 'tools', 'debug', 'stress_tail.py')
    outs = []
    for lib in (None, vlib):
        env = dict(os.environ)
        env.pop('LAFF_HIP_LIB', None)
        if lib:
            env['LAFF_HIP_LIB'] = lib
        r = subprocess.run([sys.executable, script, '150', mode], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    for o in outs:
        assert len(o['distinct']) == 1 and o['distinct'][0][1] == 150, o
        assert o['distinct'][0][0][7] == '0.0'
    assert outs[0]['distinct'][0][0] == outs[1]['distinct'][0][0] and outs[0]['rank_sum'] == outs[1]['rank_sum']
    assert outs[1]['lib'] == vlib


def test_c_abi_collectives_on_a_one_rank_group():
    """laff_comm_* / laff_allgather_rows / laff_allreduce_* (include/laff_hip.h, section e) on a 1-rank RCCL communicator: RCCL is found at
    run time, the calls run on torch's stream, and with one rank every collective is the identity."""
    import ctypes as C
    from laff_amd import _lib, ops
    lib, h = ops._context(torch.device(DEV))
    uid = (C.c_ubyte * 128)()
    _lib.check(lib.laff_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(lib.laff_comm_init(h, 0, 1, uid, C.byref(comm)))
    try:
        _lib.check(lib.laff_comm_set_stream(comm, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        g = torch.Generator(device=DEV).manual_seed(1)
        rows = torch.randn(300, 512, generator=g, device=DEV)
        out = torch.empty_like(rows)
        _lib.check(lib.laff_allgather_rows(comm, C.c_void_p(rows.data_ptr()), C.c_void_p(out.data_ptr()), rows.numel() * 4))
        cnt = torch.arange(-5, 995, device=DEV, dtype=torch.int32)
        cnt0 = cnt.clone()
        _lib.check(lib.laff_allreduce_i32_sum(comm, C.c_void_p(cnt.data_ptr()), cnt.numel()))
        sg = torch.randn(1000, generator=g, device=DEV, dtype=torch.float64)
        sg[::7] = float('-inf')
        sg0 = sg.clone()
        _lib.check(lib.laff_allreduce_f64_max(comm, C.c_void_p(sg.data_ptr()), sg.numel()))
        torch.cuda.synchronize()
        assert torch.equal(out, rows) and torch.equal(cnt, cnt0) and torch.equal(sg, sg0)
        assert lib.laff_allgather_rows(comm, None, None, 16) != 0 and b'null buffer' in lib.laff_last_error()
    finally:
        _lib.check(lib.laff_comm_destroy(comm))


def _empty_list_state(count):
    """A RankState whose pair list is empty (tiled format: header {0, 0, 0, 4}): laff_rank_resolve_metrics then reduces to the metrics
    of count + base -- the tail of the fused launch on arbitrary rank distributions."""
    from laff_amd import ops
    n = count.numel()
    d = torch.device(DEV)
    Et = torch.zeros((n, 1, 4), device=d)
    Ev = torch.zeros((1, 1, 4), device=d)
    pairs = torch.zeros(4 + 2 * 8, dtype=torch.int32, device=d)
    pairs[3] = 4
    return ops.RankState(Et, Ev, None, None, 1, None, 0, torch.zeros(n, dtype=torch.float64, device=d), None, None, count, pairs, 8)


def test_resolve_metrics_tail_on_the_median_cases():
    """The metrics computed by the last workgroup of the resolve launch (laff_rank_resolve_metrics) on the rank distributions that
    exercise every median path of the stand-alone kernel, synchronous and through a pinned buffer, ranks_out = count + base; a rank < 1
    is flagged both ways and leaves the ticket clean for the next launch."""
    from laff_amd import ops
    g = rnd(777)
    cases = [np.array([3, 3, 7, 9], np.int32), np.array([1, 1, 300, 400], np.int32), np.array([255, 256], np.int32),
             np.array([256, 255, 255, 256], np.int32), np.array([65279, 65280], np.int32), np.array([70000, 65279, 1, 65281], np.int32),
             np.array([511, 512, 512, 511], np.int32), np.array([5], np.int32), np.array([2 ** 31 - 1, 1], np.int32),
             g.integers(1, 200, 40000).astype(np.int32), g.integers(1, 200, 40001).astype(np.int32),
             g.integers(250, 262, 5000).astype(np.int32), g.integers(65270, 65290, 5003).astype(np.int32),
             g.integers(1, 50000, 300000).astype(np.int32), g.integers(1, 100, 1 << 19).astype(np.int32),
             np.concatenate([np.ones(20000, np.int32), g.integers(1, 3000, 20001).astype(np.int32)])]
    pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
    for r in cases:
        for base in (1, 0):
            count = dev(r - base, torch.int32)
            st = _empty_list_state(count)
            ranks = torch.empty_like(count)
            got = ops.rank_resolve_metrics(st, None, None, base=base, ranks_out=ranks)
            np.testing.assert_allclose(got, _metrics_numpy(r), rtol=1e-13, atol=0, err_msg=str(r[:8]))
            assert np.array_equal(ranks.cpu().numpy(), r)
            pinned.fill_(-1.0)
            assert ops.rank_resolve_metrics(st, None, pinned, base=base) is None
            torch.cuda.synchronize()
            np.testing.assert_allclose(pinned[:7].numpy(), _metrics_numpy(r), rtol=1e-13, atol=0)
            assert pinned[7].item() == 0.0
        bad = r.copy(); bad[len(bad) // 2] = 0
        st = _empty_list_state(dev(bad - 1, torch.int32))
        with pytest.raises(RuntimeError):
            ops.rank_resolve_metrics(st)
        ops.rank_resolve_metrics(st, None, pinned)
        torch.cuda.synchronize()
        assert pinned[7].item() == 1.0 and np.isnan(pinned[0].item())


@pytest.mark.parametrize('name,prec', [('tiny', 'fp16'), ('c2_10kx3k', 'fp16'), ('c2_10kx3k', 'bf16'), ('c4_40kx10k', 'fp16')])
def test_fused_tail_equals_resolve_then_metrics(name, prec, monkeypatch):
    """evaluate_sharded on one rank: GEMM + laff_rank_resolve_metrics (default) against GEMM + laff_rank_resolve + laff_rank_metrics
    (LAFF_FUSED_TAIL=0): the same ranks, the same patched score matrix, metrics equal to rounding -- both list formats (the strip
    kernel's dumps at 40k x 10k, the tiled kernel's pairs below), eager and replayed from a captured graph."""
    from laff_amd import synth
    from laff_amd.dist import HipBackend, check_metrics_flag, evaluate_sharded
    Nt, Nv, H, d, _ = synth.WORKLOADS[name]
    devc = torch.device(DEV)
    model = synth.build_model(H, d, devc)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, devc)
    be = HipBackend(model, prec)
    monkeypatch.setenv('LAFF_FUSED_TAIL', '0')
    ref = evaluate_sharded(be, vis, txt, gt, Nt, Nv, H)
    monkeypatch.setenv('LAFF_FUSED_TAIL', '1')
    got = evaluate_sharded(be, vis, txt, gt, Nt, Nv, H)
    assert torch.equal(got['ranks'], ref['ranks'])
    assert torch.equal(got['S_local'], ref['S_local'])
    np.testing.assert_allclose(got['metrics'], ref['metrics'], rtol=1e-13, atol=0)
    pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        out = evaluate_sharded(be, vis, txt, gt, Nt, Nv, H, metrics_out=pinned)
    for _ in range(3):
        pinned.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        check_metrics_flag(pinned)
        assert tuple(pinned[:7].tolist()) == tuple(got['metrics'])
        assert torch.equal(out['ranks'], ref['ranks'])


@pytest.mark.parametrize('prec', ['fp16', 'bf16'])
@pytest.mark.parametrize('Nt,Nv,H,d', [(1000, 333, 1, 512), (257, 64, 8, 64), (5000, 1250, 1, 512)])
def test_rank_prepare_emit_equals_pack_then_prepare(Nt, Nv, H, d, prec):
    """laff_rank_prepare_emit (ops.rank_prepare with T or V None) produces the operand of unit-norm rows itself: bit for bit the buffer
    pack_rows(E, normalize=False) writes, and the state (exact ground-truth scores, both bands, cleared accumulators) of the plain
    laff_rank_prepare on that operand -- for the text side, the video side and both."""
    from laff_amd import ops
    g = rnd(Nt + Nv + H)
    te = g.normal(0, 1, (Nt, H, d)).astype(np.float32)
    ve = g.normal(0, 1, (Nv, H, d)).astype(np.float32)
    te /= np.linalg.norm(te, axis=2, keepdims=True)
    ve /= np.linalg.norm(ve, axis=2, keepdims=True)
    Et, Ev = dev(te), dev(ve)
    gt = dev(g.integers(0, Nv, Nt).astype(np.int32), torch.int32)
    T = ops.pack_rows(Et, False, 1e-13, prec)
    V = ops.pack_rows(Ev, False, 1e-13, prec)
    ref = ops.rank_prepare(Et, Ev, T, V, gt)
    for t_in, v_in in ((None, V), (T, None), (None, None)):
        st = ops.rank_prepare(Et, Ev, t_in, v_in, gt, emit_precision=prec)
        assert torch.equal(st.T.buf[:Nt * H * d * 2], T.buf[:Nt * H * d * 2]) and torch.equal(st.V.buf[:Nv * H * d * 2], V.buf[:Nv * H * d * 2])
        assert st.T.precision == prec and st.V.prescale == T.prescale
        assert torch.equal(st.s_gt64, ref.s_gt64)
        assert torch.equal(st.band_t[:Nt], ref.band_t[:Nt])
        nb = ((Nv + 3) & ~3) + (Nv + 63) // 64
        assert torch.equal(st.band_v[:Nv], ref.band_v[:Nv]) and torch.equal(st.band_v[(Nv + 3) & ~3:nb], ref.band_v[(Nv + 3) & ~3:nb])
        assert int(st.count.abs().sum()) == 0 and st.pairs[:2].tolist() == [0, 0]
    with pytest.raises(ValueError):
        ops.rank_prepare(Et, Ev, None, None, gt, emit_precision='fp16x3')


def test_c_host_runs_the_exact_rank_tail(tmp_path):
    """A host without Python or torch (tests/c_host/laff_host.c: gcc, hipMalloc'ed buffers, the C ABI only) runs prepare (producing both
    operands) -> banded GEMM -> resolve + metrics: ranks equal to the ranks of float64 scores of the same embeddings, evaluation.eval's
    numbers equal to the oracle's, scores within the fp16 contract."""
    import os
    import subprocess
    import sys
    from oracle import laff_oracle as O
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_cabi import _build_c_host
    exe = _build_c_host(tmp_path)
    g = rnd(2024)
    Nt, Nv, H, d = 3000, 700, 2, 128
    z = g.normal(0, 1, (Nv, H, d)).astype(np.float32)
    gt = (np.arange(Nt) % Nv).astype(np.int32)
    te = (z[gt] + 5.0 * g.normal(0, 1, (Nt, H, d))).astype(np.float32)
    ve = z.copy()
    te /= np.linalg.norm(te, axis=2, keepdims=True)
    ve /= np.linalg.norm(ve, axis=2, keepdims=True)
    prob, out = str(tmp_path / 'problem.bin'), str(tmp_path / 'out.bin')
    with open(prob, 'wb') as f:
        np.array([Nt, Nv, H, d], np.int32).tofile(f)
        te.tofile(f); ve.tofile(f); gt.tofile(f)
    r = subprocess.run([exe, prob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-500:]
    raw = open(out, 'rb').read()
    ranks = np.frombuffer(raw, np.int32, Nt, 0)
    metrics = np.frombuffer(raw, np.float64, 7, 4 * Nt)
    S = np.frombuffer(raw, np.float32, Nt * Nv, 4 * Nt + 56).reshape(Nt, Nv)
    S64 = O.txt2vis_matrix_f64(te, ve)
    want = O.count_ranks(S64, gt)
    assert np.array_equal(ranks, want)
    assert len(set(ranks.tolist())) > 8
    np.testing.assert_allclose(metrics, O.eval_from_positions([[x] for x in want.astype(np.float64)]), rtol=1e-12, atol=0)
    assert float(np.abs(S - S64).max()) <= 1e-4
