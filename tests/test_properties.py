"""Property tests (hypothesis): invariants of the host logic / oracle on the CPU, randomised shapes of the kernels on the GPU."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

from oracle import laff_oracle as O

# derandomize: the same examples on every run (the driver's round-end run must not depend on a lucky draw)
COMMON = dict(deadline=None, derandomize=True, database=None,
              suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


# ------------------------------------------------------------------------------------------------ CPU
@settings(max_examples=200, **COMMON)
@given(n=st.integers(0, 10 ** 6), world=st.integers(1, 64))
def test_shard_bounds_partition_the_range(n, world):
    from laff_amd.dist import shard_bounds
    cuts = [shard_bounds(n, world, r) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    sizes = [hi - lo for lo, hi in cuts]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


@settings(max_examples=60, **COMMON)
@given(seed=st.integers(0, 10 ** 6), nt=st.integers(1, 40), nv=st.integers(1, 60))
def test_count_rank_equals_argsort_position_without_ties(seed, nt, nv):
    """rank = 1 + #{c != gt : s > s_gt} is the reference's argsort position (predictor.py:232-243) whenever scores are distinct."""
    g = np.random.default_rng(seed)
    S = g.permutation(nt * nv).reshape(nt, nv).astype(np.float32)          # all scores distinct
    gt = g.integers(0, nv, nt)
    inds = np.argsort(S, axis=1)
    for i in range(nt):
        order = inds[i][::-1]
        pos = int(np.where(order == gt[i])[0][0]) + 1
        assert pos == 1 + int(np.sum(np.delete(S[i], gt[i]) > S[i, gt[i]]))
        assert O.gt_positions(S[i], [gt[i]]) == [pos]


@settings(max_examples=100, **COMMON)
@given(text=st.text(alphabet=st.characters(min_codepoint=32, max_codepoint=126), max_size=80))
def test_tokenizer_is_idempotent_and_lowercase_alnum(text):
    from laff_amd import txt2vec as T
    toks = T.tokenize(text)
    assert toks == O.tokenize(text)
    assert T.tokenize(' '.join(toks)) == toks
    assert all(t and t == t.lower() and t.isalnum() and t.isascii() for t in toks)


@settings(max_examples=25, **COMMON)
@given(seed=st.integers(0, 10 ** 6), B=st.integers(2, 7), H=st.integers(1, 3), d=st.integers(2, 6),
       maxv=st.booleans(), mean=st.booleans(), direction=st.sampled_from(['i2t', 't2i', 'bidir']))
def test_margin_loss_oracle_gradient_matches_finite_differences(seed, B, H, d, maxv, mean, direction):
    """The analytic backward of the oracle's margin ranking loss against central differences in float64."""
    g = np.random.default_rng(seed)
    s = g.normal(0, 1, (B, H, d))
    im = g.normal(0, 1, (B, H, d))
    style = 'mean' if mean else 'sum'

    def f64(s_, im_):          # float64 restatement of the forward only
        total = 0.0
        for h in range(H):
            hs = s_[:, h] / np.linalg.norm(s_[:, h], axis=1, keepdims=True)
            hi = im_[:, h] / np.linalg.norm(im_[:, h], axis=1, keepdims=True)
            sc = hi @ hs.T
            dg = np.diag(sc)
            off = ~np.eye(B, dtype=bool)
            for on, cost in (('i2t', np.maximum(0.2 + sc - dg[:, None], 0) * off), ('t2i', np.maximum(0.2 + sc - dg[None, :], 0) * off)):
                if direction not in (on, 'bidir'):
                    continue
                c = cost.max(1 if on == 'i2t' else 0) if maxv else cost
                total += c.mean() if mean else c.sum()
        return total

    loss, d_s, d_im = O.margin_ranking_loss(s.astype(np.float32), im.astype(np.float32), 0.2, maxv, style, direction)
    assert abs(float(loss) - f64(s, im)) <= 1e-4 * max(1.0, abs(f64(s, im)))
    eps = 1e-6
    for _ in range(6):
        idx = tuple(int(g.integers(0, n)) for n in (B, H, d))
        for arr, grad in ((s, d_s), (im, d_im)):
            a, b = arr.copy(), arr.copy()
            a[idx] += eps
            b[idx] -= eps
            num = (f64(a, im) - f64(b, im)) / (2 * eps) if arr is s else (f64(s, a) - f64(s, b)) / (2 * eps)
            # hinge / arg-max kinks inside +-eps are measure-zero for random inputs; allow the fp32 noise of the analytic side
            assert abs(num - float(grad[idx])) <= 2e-3 * max(1.0, abs(num)), (idx, num, float(grad[idx]))


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


def _dev(a, dtype=None):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype or torch.float32, device='cuda')


@gpu
@settings(max_examples=25, **COMMON)
@given(seed=st.integers(0, 10 ** 6), N=st.integers(1, 300), L=st.integers(1, 8), H=st.sampled_from([1, 2, 4, 8]),
       dq=st.integers(1, 32), with_ave=st.booleans(), mul=st.booleans())
def test_fuse_random_shapes_vs_oracle(seed, N, L, H, dq, with_ave, mul):
    from laff_amd import ops
    d = 4 * dq
    g = np.random.default_rng(seed)
    planes = [np.tanh(g.normal(0, 1, (N, H * d))).astype(np.float32) for _ in range(L)]
    w = g.normal(0, 0.3, (H, d)).astype(np.float32)
    b = g.normal(0, 0.3, H).astype(np.float32)
    gw = g.uniform(0, 1, H).astype(np.float32)
    E = ops.fuse([(_dev(p), False, None, None) for p in planes], H, d, _dev(w), _dev(b), _dev(gw), ops.attention_flags(with_ave, mul))
    ref = O.multi_head_attention(np.stack(planes, axis=1), w, b, gw, H, with_ave, mul)
    assert np.abs(E.cpu().numpy() - ref).max() <= 5e-6


@gpu
@settings(max_examples=25, **COMMON)
@given(seed=st.integers(0, 10 ** 6), Nt=st.integers(1, 400), Nv=st.integers(1, 400), H=st.sampled_from([1, 2, 8]),
       dq=st.integers(1, 24), precision=st.sampled_from(['fp32', 'fp16', 'fp16x3', 'bf16x3']))
def test_similarity_random_shapes_vs_oracle(seed, Nt, Nv, H, dq, precision):
    from laff_amd import ops
    d = 8 * dq
    g = np.random.default_rng(seed)
    t = g.normal(0, 1, (Nt, H, d)).astype(np.float32)
    v = g.normal(0, 3, (Nv, H, d)).astype(np.float32)
    S = ops.sim_gemm(ops.pack_rows(_dev(t), True, 1e-13, precision), ops.pack_rows(_dev(v), True, 1e-13, precision), heads=H)
    ref = O.txt2vis_matrix(t, v)
    # operand rounding is relative to the element size ~ 1/sqrt(d) (fp16: 11 bits, bf16 hi+lo: 16 bits)
    tol = {'fp32': 2e-6, 'fp16x3': 2e-6, 'bf16x3': 4e-5 / np.sqrt(d) + 3e-6, 'fp16': 1e-3 / np.sqrt(d) + 1e-4}[precision]
    assert np.abs(S.cpu().numpy() - ref).max() <= tol


@gpu
@settings(max_examples=25, **COMMON)
@given(seed=st.integers(0, 10 ** 6), N=st.integers(1, 300), Dk=st.integers(1, 700), D=st.integers(1, 300),
       act=st.sampled_from([None, 'tanh', 'relu', 'sigmoid']), bn=st.booleans())
def test_fc_random_shapes_vs_oracle(seed, N, Dk, D, act, bn):
    from laff_amd import ops
    g = np.random.default_rng(seed)
    x = g.normal(0, 1, (N, Dk)).astype(np.float32)
    W = (g.normal(0, 1, (D, Dk)) / np.sqrt(Dk)).astype(np.float32)
    b = g.normal(0, 0.1, D).astype(np.float32)
    sc = g.uniform(0.5, 1.5, D).astype(np.float32) if bn else None
    sh = g.normal(0, 0.1, D).astype(np.float32) if bn else None
    y = ops.fc_act_bn(_dev(x), _dev(W), _dev(b), _dev(sc) if bn else None, _dev(sh) if bn else None, act)
    ref = O.activation((x.astype(np.float64) @ W.astype(np.float64).T + b).astype(np.float32), act)
    if bn:
        ref = ref * sc + sh
    assert np.abs(y.cpu().numpy() - ref).max() <= 2e-5


@gpu
@settings(max_examples=20, **COMMON)
@given(seed=st.integers(0, 10 ** 6), Nt=st.integers(1, 300), Nv=st.integers(1, 200), H=st.sampled_from([1, 2, 8]), dq=st.integers(1, 40),
       precision=st.sampled_from(['fp16', 'fp16x3', 'bf16']))
def test_fused_ranking_random_shapes_is_self_consistent(seed, Nt, Nv, H, dq, precision):
    """retrieval.evaluate's fused path (row_dot_gt -> GEMM epilogue count) on odd shapes: the ranks it reports are exactly the
    ranks recounted from the score matrix it returns, and the scores agree with the oracle."""
    import torch
    from laff_amd import ops
    d = 4 * dq
    g = np.random.default_rng(seed)
    t = g.normal(0, 1, (Nt, H, d)).astype(np.float32)
    v = g.normal(0, 1, (Nv, H, d)).astype(np.float32)
    gt = g.integers(0, Nv, Nt).astype(np.int32)
    T, V = ops.pack_rows(_dev(t), True, 1e-13, precision), ops.pack_rows(_dev(v), True, 1e-13, precision)
    gtd = _dev(gt, torch.int32)
    s_gt = ops.row_dot_gt(T, V, gtd, H)
    count = torch.zeros(Nt, dtype=torch.int32, device='cuda')
    S = ops.sim_gemm(T, V, heads=H, gt_col=gtd, s_gt=s_gt, count=count).cpu().numpy()
    recount = np.array([np.sum(np.delete(S[i], gt[i]) > S[i, gt[i]]) for i in range(Nt)])
    assert np.array_equal(count.cpu().numpy(), recount)
    assert np.array_equal(S[np.arange(Nt), gt], s_gt.cpu().numpy())
    ref = O.txt2vis_matrix(t, v)
    # 16-bit operand rounding is relative to the element size ~ 1/sqrt(d): the 1e-4 contract is quoted at d = 512
    h = 1e-3 / np.sqrt(d) + 1e-4
    assert np.abs(S - ref).max() <= {'fp16': h, 'fp16x3': 2e-6, 'bf16': 8 * h}[precision]


@gpu
@settings(max_examples=20, **COMMON)
@given(seed=st.integers(0, 10 ** 6), Nt=st.integers(1, 300), Nv=st.integers(1, 200), H=st.sampled_from([1, 2, 8]), dq=st.integers(1, 40),
       precision=st.sampled_from(['fp16', 'fp16x3', 'bf16']))
def test_exact_ranking_random_shapes(seed, Nt, Nv, H, dq, precision):
    """The exact-rank pipeline (prepare -> banded GEMM epilogue -> resolve) on odd shapes: its ranks are the ranks of the oracle's
    float64 scores whatever the operand precision, they are exactly the ranks recounted from the score matrix it returns, and
    the scores agree with the oracle."""
    import torch
    from laff_amd import ops
    d = 4 * dq
    g = np.random.default_rng(seed)
    t = g.normal(0, 1, (Nt, H, d)).astype(np.float32)
    v = g.normal(0, 1, (Nv, H, d)).astype(np.float32)
    gt = g.integers(0, Nv, Nt).astype(np.int32)
    Et, Ev = _dev(t), _dev(v)
    T, V = ops.pack_rows(Et, True, 1e-13, precision), ops.pack_rows(Ev, True, 1e-13, precision)
    gtd = _dev(gt, torch.int32)
    S, count, state = ops.exact_ranks(Et, Ev, T, V, gtd)
    S = S.cpu().numpy()
    assert np.array_equal(count.cpu().numpy() + 1, O.count_ranks(O.txt2vis_matrix_f64(t, v), gt))
    recount = np.array([np.sum(np.delete(S[i], gt[i]) > S[i, gt[i]]) for i in range(Nt)])
    assert np.array_equal(count.cpu().numpy(), recount)
    assert np.array_equal(S[np.arange(Nt), gt], state.s_gt64.float().cpu().numpy())
    ref = O.txt2vis_matrix(t, v)
    # 16-bit operand rounding is relative to the element size ~ 1/sqrt(d): the 1e-4 contract is quoted at d = 512
    h = 1e-3 / np.sqrt(d) + 1e-4
    assert np.abs(S - ref).max() <= {'fp16': h, 'fp16x3': 2e-6, 'bf16': 8 * h}[precision]


@gpu
@settings(max_examples=20, **COMMON)
@given(seed=st.integers(0, 10 ** 6), B=st.integers(1, 60), Fmax=st.integers(1, 40), dq=st.integers(1, 64), with_ave=st.booleans(),
       mul=st.booleans(), use_lens=st.booleans())
def test_frame_attention_random_shapes_vs_oracle(seed, B, Fmax, dq, with_ave, mul, use_lens):
    import torch
    from laff_amd import ops
    d = 4 * dq
    g = np.random.default_rng(seed)
    lens = g.integers(0, Fmax + 1, B).astype(np.int32)
    frames = np.zeros((B, Fmax, d), np.float32)
    for i in range(B):
        frames[i, :lens[i]] = g.normal(0, 1, (lens[i], d))
    w = (g.uniform(-1, 1, d) / np.sqrt(d)).astype(np.float32)
    b, gw = np.float32(g.normal(0, 0.3)), np.float32(g.uniform(0, 1))
    got = ops.frame_fuse(_dev(frames), _dev(lens, torch.int32) if use_lens else None, _dev(w), _dev(b).view(1), _dev(gw).view(1),
                         ops.attention_flags(with_ave, mul)).cpu().numpy()
    ref = O.frame_attention(frames, w, b, with_ave, mul, gw)
    m = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), m)
    if m.any():
        assert np.abs(got[m] - ref[m]).max() <= 3e-6


@gpu
@settings(max_examples=20, **COMMON)
@given(seed=st.integers(0, 10 ** 6), Nt=st.integers(1, 40), Nv=st.integers(1, 3000), K=st.integers(1, 600))
def test_topk_random_shapes_vs_stable_argsort(seed, Nt, Nv, K):
    from laff_amd import ops
    g = np.random.default_rng(seed)
    S = g.normal(0, 1, (Nt, Nv)).astype(np.float32)
    S[:, : Nv // 3] = np.round(S[:, : Nv // 3], 1)            # plenty of exact ties
    k = min(K, Nv)
    idx, val = ops.topk_rows(_dev(S), k)
    ref = np.argsort(S, axis=1, kind='stable')[:, ::-1][:, :k]
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert np.array_equal(val.cpu().numpy(), np.take_along_axis(S, ref, axis=1))


@gpu
@settings(max_examples=15, **COMMON)
@given(seed=st.integers(0, 10 ** 6), B=st.integers(1, 200), H=st.sampled_from([1, 2, 8]), dq=st.integers(1, 40), maxv=st.booleans(),
       mean=st.booleans(), direction=st.sampled_from(['i2t', 't2i', 'bidir']))
def test_margin_loss_random_shapes_vs_oracle(seed, B, H, dq, maxv, mean, direction):
    from laff_amd import ops
    d = 4 * dq
    g = np.random.default_rng(seed)
    z = g.normal(0, 1, (B, 8)).astype(np.float32)
    P = g.normal(0, 1, (8, H * d)).astype(np.float32)
    s = (z @ P + 2.0 * g.normal(0, 1, (B, H * d))).astype(np.float32).reshape(B, H, d)
    im = (z @ P + 2.0 * g.normal(0, 1, (B, H * d))).astype(np.float32).reshape(B, H, d)
    style = 'mean' if mean else 'sum'
    loss, d_s, d_im = ops.margin_loss(_dev(s), _dev(im), 0.2, maxv, style, direction)
    rl, rs, ri = O.margin_ranking_loss(s, im, 0.2, maxv, style, direction)
    assert abs(loss.item() - float(rl)) <= 1e-4 * max(1.0, abs(float(rl)))
    # a decision flipped by a 1-ulp score difference moves whole gradient rows: robust comparison
    bad = max((np.abs(d_s.cpu().numpy() - rs).max(axis=-1) > 1e-5).mean(), (np.abs(d_im.cpu().numpy() - ri).max(axis=-1) > 1e-5).mean())
    assert bad <= 0.03


@gpu
@settings(max_examples=20, **COMMON)
@given(seed=st.integers(0, 10 ** 6), nv=st.integers(1, 60), max_caps=st.integers(1, 25))
def test_both_retrieval_directions_random_groupings_vs_oracle(seed, nv, max_caps):
    """predictor.py:232-270 on the device for an arbitrary number of captions per video (1 .. max_caps, some videos with many):
    T2V and V2T metrics equal the argsort / label-matrix restatement when scores are distinct."""
    from laff_amd import predictor as P
    g = np.random.default_rng(seed)
    caps = g.integers(1, max_caps + 1, nv)
    txt_ids, vis_ids = [], ['vid%d' % v for v in range(nv)]
    for v in g.permutation(nv):
        txt_ids += ['vid%d#%d' % (v, c) for c in range(caps[v])]
    nt = len(txt_ids)
    S = (g.permutation(nt * nv).reshape(nt, nv) / float(nt * nv)).astype(np.float32)         # all distinct
    t2v, v2t = P.retrieval_metrics(S, txt_ids, vis_ids)
    rt, rv = O.predictor_metrics(S, txt_ids, vis_ids)
    np.testing.assert_allclose(t2v, rt, rtol=1e-12)
    np.testing.assert_allclose(v2t, rv, rtol=1e-12)
