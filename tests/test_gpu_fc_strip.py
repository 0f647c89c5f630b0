"""Strip form of TransformNet.forward (laff_fc_act_bn_strip_grouped; /root/reference/model/model.py:257-276) on the GPU: parity with
the oracle / float64, with the fp32-MFMA path and with the tiled fp16x3 path, through the C ABI."""
import numpy as np
import pytest
import torch

from tests.util import maxdiff

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def rnd(seed):
    return np.random.default_rng(seed)


def _layer(g, D, bias=True, bn=True):
    W = (g.normal(0, 1, (D, 512)) / np.sqrt(512)).astype(np.float32)
    b = g.normal(0, 0.1, D).astype(np.float32) if bias else None
    sc = g.uniform(0.5, 1.5, D).astype(np.float32) if bn else None
    sh = g.normal(0, 0.1, D).astype(np.float32) if bn else None
    return W, b, sc, sh


def _ref64(x, W, b, sc, sh, act):
    pre = x.astype(np.float64) @ W.astype(np.float64).T + (0.0 if b is None else b)
    f = {None: lambda v: v, 'tanh': np.tanh, 'relu': lambda v: np.maximum(v, 0), 'sigmoid': lambda v: 1 / (1 + np.exp(-np.clip(v, -700, 700)))}[act]
    ref = f(pre) * (1.0 if sc is None else sc) + (0.0 if sh is None else sh)
    scale = 1.0 if act in ('tanh', 'sigmoid') else np.maximum(1.0, np.abs(pre).max(axis=1, keepdims=True))
    return ref, scale


@pytest.mark.parametrize('N,D,act', [(1, 32, 'tanh'), (31, 64, 'tanh'), (128, 512, 'tanh'), (129, 512, None), (300, 512, 'relu'),
                                      (1000, 512, 'sigmoid'), (257, 4096, 'tanh'), (5000, 512, 'tanh'), (4097, 96, None)])
def test_fc_strip_vs_fp64(N, D, act):
    """Same tolerance as the other FC forms (test_fc_split_fp16x3_vs_fp64), incl. a row beyond the fp16 range, a tiny row and a zero
    row -- the per-row scale is found in registers by the kernel itself."""
    from laff_amd import ops
    g = rnd(N + D)
    x = g.normal(0, 1, (N, 512)).astype(np.float32)
    x[0] *= 3e4
    if N > 2:
        x[1] *= 1e-6
        x[2] = 0
    W, b, sc, sh = _layer(g, D)
    sw = ops.fc_strip_pack(dev(W), dev(b), dev(sc), dev(sh), act)
    y = ops.fc_act_bn_strip_grouped([dict(x=dev(x), strip=sw)])[0]
    ref, scale = _ref64(x, W, b, sc, sh, act)
    assert float(np.max(np.abs(y.cpu().numpy() - ref) / scale)) <= 2e-5
    y32 = ops.fc_act_bn(dev(x), dev(W), dev(b), dev(sc), dev(sh), act)
    assert float(np.max(np.abs((y - y32).cpu().numpy()) / scale)) <= 2e-5


@pytest.mark.parametrize('N,D', [(40000, 64), (40000, 96), (20000, 128), (33000, 160), (9000, 32), (70000, 32), (12800, 1024)])
def test_fc_strip_segment_shapes(N, D):
    """A workgroup's range is cut into segments (the blocks of one strip).  Segments of 1, 2, 3, ... blocks with and without a successor
    take different code: the plain loop may be empty, the last two blocks are the copies whose bodies fetch the next strip's first rounds,
    one-block segments fetch nothing ahead.  Narrow outputs over many rows produce all of them."""
    from laff_amd import ops
    g = rnd(N + D)
    x = g.normal(0, 1, (N, 512)).astype(np.float32)
    W, b, sc, sh = _layer(g, D)
    sw = ops.fc_strip_pack(dev(W), dev(b), dev(sc), dev(sh), 'tanh')
    y = ops.fc_act_bn_strip_grouped([dict(x=dev(x), strip=sw)])[0]
    y32 = ops.fc_act_bn(dev(x), dev(W), dev(b), dev(sc), dev(sh), 'tanh')
    assert float((y - y32).abs().max()) <= 2e-5
    rows = np.r_[0:300, N - 300:N]
    ref, _ = _ref64(x[rows], W, b, sc, sh, 'tanh')
    assert float(np.max(np.abs(y.cpu().numpy()[rows] - ref))) <= 2e-5


@pytest.mark.parametrize('bias,bn', [(False, False), (True, False), (False, True)])
def test_fc_strip_optional_stages_and_padded_rows(bias, bn):
    """bias / BatchNorm absent; input and output rows with a pitch (views of wider buffers): nothing outside [N, D] is written."""
    from laff_amd import ops
    g = rnd(17 + bias + 2 * bn)
    N, D = 333, 160
    x = g.normal(0, 1, (N, 512)).astype(np.float32)
    W, b, sc, sh = _layer(g, D, bias, bn)
    xb = torch.zeros((N, 520), device='cuda')
    xb[:, :512] = dev(x)
    out_full = torch.full((N, D + 5), -7.0, device='cuda')
    sw = ops.fc_strip_pack(dev(W), None if b is None else dev(b), None if sc is None else dev(sc), None if sh is None else dev(sh), 'tanh')
    y = ops.fc_act_bn_strip_grouped([dict(x=xb[:, :512], strip=sw, out=out_full[:, :D])])[0]
    ref, _ = _ref64(x, W, b, sc, sh, 'tanh')
    assert maxdiff(y, ref.astype(np.float32)) <= 5e-6
    assert bool((out_full[:, D:] == -7.0).all())


def test_fc_strip_grouped_mixed_launch():
    """Eight problems of different sizes, two output widths and two activation kinds in one call: the library sorts them into
    launches by (D, kind); empty problems are legal."""
    from laff_amd import ops
    g = rnd(5)
    probs, refs = [], []
    for i, (N, D, act) in enumerate([(1000, 512, 'tanh'), (0, 512, 'tanh'), (777, 512, 'sigmoid'), (130, 64, 'tanh'), (250, 512, None),
                                     (250, 512, 'relu'), (31, 64, 'tanh'), (2050, 512, 'tanh')]):
        x = g.normal(0, 1, (N, 512)).astype(np.float32)
        W, b, sc, sh = _layer(g, D)
        probs.append(dict(x=dev(x), strip=ops.fc_strip_pack(dev(W), dev(b), dev(sc), dev(sh), act)))
        refs.append(_ref64(x, W, b, sc, sh, act))
    ys = ops.fc_act_bn_strip_grouped(probs)
    for y, (ref, scale) in zip(ys, refs):
        assert y.shape == ref.shape
        if ref.size:
            assert float(np.max(np.abs(y.cpu().numpy() - ref) / scale)) <= 2e-5


def test_fc_strip_matches_tiled_fp16x3_at_the_c4_group():
    """The C4 projection group (4 x 40,000 + 4 x 10,000 rows of 512 -> 512): strip form vs the tiled fused-split form, whole
    outputs; and a float64 check on sampled rows."""
    from laff_amd import ops
    g = rnd(1)
    ps, pf, keep = [], [], []
    for N in (40000,) * 4 + (10000,) * 4:
        x = g.normal(0, 1, (N, 512)).astype(np.float32)
        W, b, sc, sh = _layer(g, 512)
        xd, Wd, bd, scd, shd = dev(x), dev(W), dev(b), dev(sc), dev(sh)
        ps.append(dict(x=xd, strip=ops.fc_strip_pack(Wd, bd, scd, shd, 'tanh')))
        pf.append(dict(x=xd, weight_split=ops.split_rows(Wd), bias=bd, bn_scale=scd, bn_shift=shd, activation='tanh'))
        keep.append((x, W, b, sc, sh))
    ys = ops.fc_act_bn_strip_grouped(ps)
    yf = ops.fc_act_bn_fused_grouped(pf)
    for a, b_ in zip(ys, yf):
        assert maxdiff(a, b_) <= 5e-6
    rows = g.integers(0, 10000, 64)
    for y, (x, W, b, sc, sh) in zip(ys, keep):
        ref, _ = _ref64(x[rows], W, b, sc, sh, 'tanh')
        assert float(np.abs(y[torch.from_numpy(rows).cuda()].cpu().numpy() - ref).max()) <= 5e-6


def test_fc_strip_refuses_what_it_cannot_take():
    from laff_amd import ops
    g = rnd(3)
    W, b, sc, sh = _layer(g, 64)
    with pytest.raises(RuntimeError):
        ops.fc_strip_pack(dev(W[:, :256].copy()), dev(b), dev(sc), dev(sh), 'tanh')          # Dk != 512
    with pytest.raises(RuntimeError):
        ops.fc_strip_pack(dev(W[:40]), dev(b[:40]), dev(sc[:40]), dev(sh[:40]), 'tanh')      # D % 32 != 0
    sw = ops.fc_strip_pack(dev(W), dev(b), dev(sc), dev(sh), 'tanh')
    with pytest.raises(ValueError):
        ops.fc_act_bn_strip_grouped([dict(x=dev(g.normal(0, 1, (8, 256)).astype(np.float32)), strip=sw)])
    with pytest.raises(RuntimeError):
        ops.fc_act_bn_strip_grouped([dict(x=torch.zeros((8, 512)), strip=sw)])               # CPU tensor: no CPU path


def test_towers_take_the_strip_form_and_keep_the_tolerance(monkeypatch):
    """'LAFF' towers with FC_PRECISION = 'fp16x3': the 512-d projections go through laff_fc_act_bn_strip_grouped (counted), and the
    embeddings stay inside the tolerance of the fp32 path; LAFF_FC_STRIP=0's tiled route gives the same embeddings."""
    from laff_amd import ops, retrieval, synth
    import laff_amd.model.model as M
    devc = torch.device('cuda')
    model = synth.build_model(1, 512, devc, seed=11)
    vis, txt, gt, _ = synth.make_features(700, 300, devc, seed=11)
    calls = []
    real = ops.fc_act_bn_strip_grouped
    monkeypatch.setattr(ops, 'fc_act_bn_strip_grouped', lambda pr: (calls.append(len(pr)), real(pr))[1])
    try:
        with torch.no_grad():
            v32, t32 = retrieval.embed(model, vis, txt)
            M.FC_PRECISION = 'fp16x3'
            vs, ts = retrieval.embed(model, vis, txt)
            assert calls and sum(calls) == 8
            monkeypatch.setattr(M, 'FC_STRIP', False)
            vt, tt = retrieval.embed(model, vis, txt)
            assert sum(calls) == 8
    finally:
        M.FC_PRECISION = 'fp32'
    assert maxdiff(vs, v32) <= 5e-6 and maxdiff(ts, t32) <= 5e-6
    assert maxdiff(vs, vt) <= 5e-6 and maxdiff(ts, tt) <= 5e-6
