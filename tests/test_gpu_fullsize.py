"""BASELINE.json-sized runs on a real MI355X, checked through size-independent properties (the oracle cannot run at
these sizes in seconds): self-consistency of the fused ranking, linearity / checksum of the score matrix, unit diagonal,
shard-sum identity, and the FrameLAFF (C3) and LAFF-ml (C5) workloads end to end."""
import numpy as np
import pytest
import torch

from oracle import laff_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _c4(precision='fp16', fc='fp16x3'):
    import laff_amd.model.model as M
    from laff_amd import retrieval, synth
    M.FC_PRECISION = fc
    try:
        Nt, Nv, H, d, _ = synth.WORKLOADS['c4_40kx10k']
        model = synth.build_model(H, d, torch.device(DEV))
        vis, txt, gt, _ = synth.make_features(Nt, Nv, torch.device(DEV))
        res = retrieval.evaluate(model, vis, txt, gt, precision=precision)
    finally:
        M.FC_PRECISION = 'fp32'
    return res, gt


def test_c4_40k_x_10k_properties():
    from laff_amd import ops
    res, gt = _c4()
    S, E_t, E_v = res.S, res.txt_emb, res.vis_emb
    assert tuple(S.shape) == (40000, 10000)
    # embeddings are unit vectors; scores are cosines
    assert float((E_t.norm(dim=-1) - 1).abs().max()) <= 2e-6 and float((E_v.norm(dim=-1) - 1).abs().max()) <= 2e-6
    assert float(S.abs().max()) <= 1 + 1e-4
    # rank recounted from the S we return == rank counted in the GEMM epilogue, exactly
    s_gt = ops.gather_gt(S, gt)
    recount = ops.rank_count(S, gt, s_gt) + 1
    assert torch.equal(recount, res.ranks)
    # a random sample of entries against fp64 dot products of the fp32 embeddings: the 1e-4 contract
    g = torch.Generator(device=DEV).manual_seed(0)
    ti = torch.randint(0, 40000, (200000,), device=DEV, generator=g)
    vi = torch.randint(0, 10000, (200000,), device=DEV, generator=g)
    ref = (E_t[ti, 0].double() * E_v[vi, 0].double()).sum(dim=1)
    assert float((S[ti, vi].double() - ref).abs().max()) <= 1e-4
    # checksum of checksums (linearity): sum_v S[t, v] == <E_t[t], sum_v E_v[v]>.  The rounding of the text row to fp16
    # (relative 2^-11 per element) is common to the whole row, so the bound scales with || sum_v E_v ||.
    vsum = E_v[:, 0].double().sum(dim=0)
    lhs = S.double().sum(dim=1)
    rhs = E_t[:, 0].double() @ vsum
    assert float((lhs - rhs).abs().max()) <= 2.0 ** -11 * float(vsum.norm()) + 1e-4 * 10000 ** 0.5
    # metrics are what numpy computes from the ranks
    r = res.ranks.cpu().numpy().astype(np.float64)
    exp = (100.0 * np.mean(r <= 1), 100.0 * np.mean(r <= 5), 100.0 * np.mean(r <= 10), np.floor(np.median(r)), r.mean(),
           (1.0 / r).mean(), (1.0 / r).mean())
    np.testing.assert_allclose(res.metrics, exp, rtol=1e-13)
    assert 5.0 < res.metrics[0] < 95.0          # the synthetic task is neither trivial nor chance


def _fp64_ranks(E_t, E_v, gt, rows=None):
    """ranks of the float64 cosine scores of fp32 embeddings (N, H, d), computed with torch on the device in row blocks."""
    Nt, H = E_t.shape[0], E_t.shape[1]
    v3 = E_v.double()
    v3 = v3 / (v3.pow(2).sum(2, keepdim=True).sqrt() + (1e-13 + 1e-14))
    idx = torch.arange(Nt, device=DEV) if rows is None else rows
    out = torch.empty(idx.numel(), dtype=torch.int32, device=DEV)
    for a in range(0, idx.numel(), 4096):
        r = idx[a:a + 4096]
        tb = E_t[r].double()
        tb = tb / (tb.pow(2).sum(2, keepdim=True).sqrt() + (1e-13 + 1e-14))
        S64 = torch.einsum('thd,vhd->tv', tb, v3) / H
        g = gt[r].long()
        ab = S64 > S64.gather(1, g[:, None])
        ab[torch.arange(ab.shape[0], device=DEV), g] = False
        out[a:a + 4096] = ab.sum(1).to(torch.int32) + 1
    return out


def test_c4_precisions_agree_on_ranks():
    """The headline workload: whatever the operand precision of the similarity GEMM, the ranks are those of the exact cosine
    scores of the fp32 embeddings (the reference ranks on fp32 scores, predictor.py:232-244) -- torch.equal, not 'almost'.
    Scores stay inside the 1e-4 contract; the 7 metrics are therefore identical too."""
    a, gt = _c4('fp16')
    want = _fp64_ranks(a.txt_emb, a.vis_emb, gt)
    assert torch.equal(a.ranks, want)
    n, overflow = a.rank_state.listed_pairs()
    assert not overflow and n < 8 * 40000                    # a handful of in-band pairs per query
    b, _ = _c4('fp16x3')
    assert torch.equal(a.txt_emb, b.txt_emb)                  # same towers
    assert torch.equal(b.ranks, want) and a.metrics == b.metrics
    assert float((a.S - b.S).abs().max()) <= 1e-4
    c, _ = _c4('bf16')
    assert torch.equal(c.ranks, want)
    # FC on the fp32 MFMA instead of the split fp16 pipe: embeddings move by ~1e-6, so a rank can only move where two videos
    # score within that distance of each other
    f, _ = _c4('fp16x3', fc='fp32')
    assert float((f.txt_emb - a.txt_emb).abs().max()) <= 2e-5
    assert float((f.ranks == a.ranks).float().mean()) >= 0.999
    assert max(abs(x - y) for x, y in zip(a.metrics[:3], f.metrics[:3])) <= 0.01 and a.metrics[3] == f.metrics[3]


def test_c4_shard_sum_identity():
    """Column shards (8 'GPUs' emulated sequentially on one): MAX of the shards' exact ground-truth scores, SUM of the shards'
    exact counts == the global ranks."""
    from laff_amd import ops
    from laff_amd.dist import shard_bounds
    res, gt = _c4()
    T = ops.pack_rows(res.txt_emb, True, 1e-13, 'fp16')
    states = []
    for r in range(8):
        v0, v1 = shard_bounds(10000, 8, r)
        Ev = res.vis_emb[v0:v1].contiguous()
        states.append(ops.rank_prepare(res.txt_emb, Ev, T, ops.pack_rows(Ev, True, 1e-13, 'fp16'), gt, col0=v0))
    s_gt = torch.stack([s.s_gt64 for s in states]).max(dim=0).values
    assert torch.isfinite(s_gt).all()
    total = torch.zeros(40000, dtype=torch.int32, device=DEV)
    for s in states:
        s.s_gt64.copy_(s_gt)
        ops.sim_gemm_banded(s, want_scores=False)
        total += ops.rank_resolve(s)
    assert torch.equal(total + 1, res.ranks)


def test_c3_framelaff_workload():
    """C3: 3k videos x 32 frames x 4 frame features, 10k texts; masked == unmasked frame attention."""
    from laff_amd import retrieval, synth
    Nt, Nv, H, d, F = synth.WORKLOADS['c3_framelaff_10kx3k']
    model = synth.build_model(H, d, torch.device(DEV), frames=F)
    vis, txt, gt, lens = synth.make_features(Nt, Nv, torch.device(DEV), frames=F)
    res = retrieval.evaluate(model, vis, txt, gt)
    assert tuple(res.S.shape) == (Nt, Nv) and tuple(res.vis_emb.shape) == (Nv, 1, 512)
    assert float((res.vis_emb.norm(dim=-1) - 1).abs().max()) <= 2e-6
    # zero-padding equivalence: a fully-masked copy (all frames marked valid) gives the same embeddings
    vis2 = dict(vis)
    vis2['mask_tensor'] = torch.ones_like(vis['mask_tensor'])
    res2 = retrieval.evaluate(model, vis2, txt, gt)
    assert float((res.vis_emb - res2.vis_emb).abs().max()) <= 2e-6
    assert float((res.S - res2.S).abs().max()) <= 1e-4
    assert float((res.ranks == res2.ranks).float().mean()) > 0.99      # embeddings differ by ~1e-7: chance-level, tie-dense ranks
    assert torch.equal(res.ranks, _fp64_ranks(res.txt_emb, res.vis_emb, gt))
    # count-only mode on chance-level scores: 1 % of the pairs sit inside the error band, the default pair list must hold them whichever
    # kernel the dispatch picks (a lower strip threshold for this mode overflowed it: 96-byte dumps instead of 8-byte pairs)
    res3 = retrieval.evaluate(model, vis, txt, gt, write_scores=False)
    assert torch.equal(res3.ranks, res.ranks) and res3.metrics == res.metrics


def test_c5_laff_ml_100k_x_30k_bf16():
    """C5: 8 heads x 512, 100k x 30k, bf16 similarity operands; ranks are still the exact ones (checked on a row sample against
    float64 scores; the 12 GB score matrix is never materialised)."""
    from laff_amd import ops, retrieval, synth
    import laff_amd.model.model as M
    M.FC_PRECISION = 'fp16x3'
    try:
        Nt, Nv, H, d, _ = synth.WORKLOADS['c5_ml_100kx30k']
        model = synth.build_model(H, d, torch.device(DEV))
        vis, txt, gt, _ = synth.make_features(Nt, Nv, torch.device(DEV))
        res = retrieval.evaluate(model, vis, txt, gt, precision='bf16', write_scores=False)
    finally:
        M.FC_PRECISION = 'fp32'
    assert res.S is None and tuple(res.txt_emb.shape) == (Nt, 8, 512)
    n, overflow = res.rank_state.listed_pairs()
    assert not overflow
    rows = torch.arange(0, Nt, 97, device=DEV)
    assert torch.equal(res.ranks[rows], _fp64_ranks(res.txt_emb, res.vis_emb, gt, rows))      # bf16 operands, exact ranks
    r = res.ranks.cpu().numpy().astype(np.float64)
    assert abs(res.metrics[0] - 100.0 * np.mean(r <= 1)) < 1e-9


def test_c4_video_to_text_positions():
    """V2T direction at 40k x 10k (4 captions per video): device counts vs numpy on a sample of videos, and the 7 metrics
    against evaluation.eval arithmetic from those positions."""
    from laff_amd import predictor
    res, gt = _c4()
    S = res.S
    owner = gt.cpu().numpy()
    pos, order, off = predictor.v2t_positions(S, owner)
    cols = np.arange(0, 10000, 397)
    Sc = S[:, torch.as_tensor(cols, device=DEV)].cpu().numpy()
    for j, v in enumerate(cols):
        texts = np.where(owner == v)[0]
        col = Sc[:, j]
        exp = np.array([1 + np.sum(col > col[t]) for t in texts])
        assert np.array_equal(pos[texts], exp)
    m = predictor.v2t_metrics(S, owner)
    first = np.array([pos[order[off[v]:off[v + 1]]].min() for v in range(10000)], dtype=np.float64)
    assert abs(m[0] - 100.0 * np.mean(first <= 1)) < 1e-9 and m[3] == np.floor(np.median(first))
    t2v = predictor.t2v_metrics(S, owner)
    np.testing.assert_allclose(t2v, res.metrics, rtol=1e-13)


def test_distributed_path_on_one_rank_rccl_group():
    """The N > 1 code path (all_gather of the text operand, all_reduce MAX / SUM, per-phase HIP graphs) on a 1-rank RCCL
    group: same ranks, scores and metrics as the plain single-GPU pass."""
    import socket
    import torch.distributed as dist
    from laff_amd import synth
    from laff_amd.dist import GraphRunner, HipBackend, evaluate_sharded
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device(DEV)
    Nt, Nv, H, d, _ = synth.WORKLOADS['c2_10kx3k']
    model = synth.build_model(H, d, dev)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev)
    backend = HipBackend(model, 'fp16')
    ref = evaluate_sharded(backend, vis, txt, gt, Nt, Nv, H)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, world_size=1, rank=0, device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        eager = evaluate_sharded(backend, vis, txt, gt, Nt, Nv, H, force_collectives=True)
        assert torch.equal(eager['ranks'], ref['ranks']) and torch.equal(eager['S_local'], ref['S_local'])
        np.testing.assert_allclose(eager['metrics'], ref['metrics'], rtol=1e-13)
        pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
        runner, state = GraphRunner(), {}
        for _ in range(3):                       # capture, then two replays
            out = evaluate_sharded(backend, vis, txt, gt, Nt, Nv, H, force_collectives=True, runner=runner, state=state,
                                   metrics_out=pinned)
            torch.cuda.synchronize()
            assert torch.equal(out['ranks'], ref['ranks']) and torch.equal(out['S_local'], ref['S_local'])
            np.testing.assert_allclose(pinned[:7].numpy(), ref['metrics'], rtol=1e-13)
            assert pinned[7].item() == 0
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------- C1: MSR-VTT test3k shapes
def _c1(name):
    from laff_amd import synth
    Nt, Nv, H, d, _ = synth.WORKLOADS[name]
    spec = synth.SPECS[name]
    dev = torch.device(DEV)
    model = synth.build_model(H, d, dev, spec=spec)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, spec=spec)
    return model, vis, txt, gt, Nt, Nv, H


def test_c1_small_whole_path_vs_oracle():
    """2+2 features, no-transform CLIP (tiled over 8 heads + BN), sparse bow through the gather FC, 20 captions per video."""
    from util import oracle_towers
    from laff_amd import retrieval, synth
    model, vis, txt, gt, Nt, Nv, H = _c1('tiny_c1')
    res = retrieval.evaluate(model, vis, txt, gt, precision='fp32')
    ve, te = oracle_towers(model, synth.to_numpy_dict(vis), synth.to_numpy_dict(txt))
    assert np.abs(res.vis_emb.cpu().numpy() - ve).max() <= 1e-5
    assert np.abs(res.txt_emb.cpu().numpy() - te).max() <= 1e-5
    S = O.txt2vis_matrix(te, ve)
    assert np.abs(res.S.cpu().numpy() - S).max() <= 1e-5
    gtn = gt.cpu().numpy()
    assert gtn.max() == Nv - 1 and np.all(np.bincount(gtn) == 20)
    from laff_amd import predictor as P
    txt_ids = ['v%d#%d' % (i // 20, i % 20) for i in range(Nt)]
    vis_ids = ['v%d' % i for i in range(Nv)]
    t2v_ref, v2t_ref = O.predictor_metrics(S, txt_ids, vis_ids)
    t2v, v2t = P.retrieval_metrics(res.S, txt_ids, vis_ids)
    assert np.allclose(t2v, res.metrics, rtol=1e-6)
    # the ranking machinery is exact: the oracle's argsort / label loop on the HIP path's OWN scores gives the same 7 + 7 metrics
    t2v_own, v2t_own = O.predictor_metrics(res.S.cpu().numpy(), txt_ids, vis_ids)
    assert np.allclose(t2v, t2v_own, rtol=0, atol=1e-9) and np.allclose(v2t, v2t_own, rtol=0, atol=1e-9)
    # against the oracle's scores (which differ by ~1e-6): a rank may move only where the oracle itself has another video within
    # 2e-5 of the ground-truth score, and by no more than the number of such videos
    want = O.count_ranks(S.astype(np.float64), gtn)
    sg = S[np.arange(Nt), gtn]
    nn = (np.abs(S - sg[:, None]) < 2e-5).sum(axis=1) - 1
    got = res.ranks.cpu().numpy()
    assert np.all(np.abs(got - want) <= nn) and np.array_equal(got[nn == 0], want[nn == 0]) and (nn == 0).mean() > 0.5
    del t2v_ref, v2t_ref


def test_c1_test3k_shapes_sparse_bow_equals_dense_and_oracle_sample():
    from util import oracle_towers
    from laff_amd import retrieval, synth
    model, vis, txt, gt, Nt, Nv, H = _c1('c1_test3k')
    assert (Nt, Nv, H) == (59800, 2990, 8)
    res = retrieval.evaluate(model, vis, txt, gt, precision='fp16')
    # the dense formulation of the same bow matrix (what the reference computes) gives the same text embeddings
    txt_dense = dict(txt, bow_encoding=txt['bow_encoding'].to_dense())
    res_d = retrieval.evaluate(model, vis, txt_dense, gt, precision='fp16')
    assert (res.txt_emb - res_d.txt_emb).abs().max().item() <= 2e-5
    assert (res.ranks == res_d.ranks).float().mean().item() >= 0.999     # (embeddings differ by ~1e-6 between the two formulations)
    # oracle on all videos and a sample of captions
    rows = np.arange(0, Nt, 299)
    ve, te = oracle_towers(model, synth.to_numpy_dict(vis), {k: v for k, v in synth.to_numpy_dict(txt).items()}, rows_t=rows)
    assert np.abs(res.vis_emb.cpu().numpy() - ve).max() <= 1e-5
    assert np.abs(res.txt_emb[torch.as_tensor(rows, device=DEV)].cpu().numpy() - te).max() <= 1e-5
    S = O.txt2vis_matrix(te, ve)
    assert np.abs(res.S[torch.as_tensor(rows, device=DEV)].cpu().numpy() - S).max() <= 1e-4
    strict = retrieval.evaluate(model, vis, txt, gt, precision='fp16x3')
    assert (strict.S - res.S).abs().max().item() <= 1e-4
    assert torch.equal(strict.ranks, res.ranks) and strict.metrics == res.metrics
    rs = torch.as_tensor(rows, device=DEV)
    assert torch.equal(res.ranks[rs], _fp64_ranks(res.txt_emb, res.vis_emb, gt, rs))
    # against the ORACLE's ranks on the sampled captions (numpy fp32 towers, float64 scores): the two sets of embeddings differ by
    # ~1e-6, so a rank may differ only where the oracle's own scores put another video within 2e-5 of the ground truth
    S64 = O.txt2vis_matrix_f64(te, ve)
    gs = gt.cpu().numpy()[rows]
    want = O.count_ranks(S64, gs)
    got = res.ranks[rs].cpu().numpy()
    sg = S64[np.arange(len(rows)), gs]
    nn = (np.abs(S64 - sg[:, None]) < 2e-5).sum(axis=1) - 1          # other videos that close to the ground truth, per row
    assert np.all(np.abs(got - want) <= nn) and (nn > 0).mean() < 0.2
    assert res.metrics[0] > 5.0                                  # far above chance (0.03 %)


def test_split_fc_with_tail_split_matches_fp32_path():
    """C4's eight projections in one grouped launch: 1,576 big tiles = 6 full rounds of 256 CUs + 40, whose last 40 tiles run as
    160 quarter tiles on the small tile body.  Same result as the fp32-MFMA path problem by problem."""
    from laff_amd import ops
    torch.manual_seed(0)
    rows = [40000] * 4 + [10000] * 4
    W = [torch.randn(512, 512, device=DEV) / 22 for _ in rows]
    Ws = [ops.split_rows(w) for w in W]
    X = [torch.randn(n, 512, device=DEV) for n in rows]
    X[5][7] *= 3e4                                   # a huge and a tiny row: the per-row scales must carry them
    X[1][11] *= 1e-6
    b = torch.randn(512, device=DEV) * 0.1
    sc = torch.rand(512, device=DEV) + 0.5
    sh = torch.randn(512, device=DEV) * 0.1
    probs = [dict(x=X[i], weight_split=Ws[i], bias=b, bn_scale=sc, bn_shift=sh, activation='tanh') for i in range(8)]
    outs = ops.fc_act_bn_split_grouped(probs)
    for i in (0, 3, 4, 5, 7):
        ref = ops.fc_act_bn(X[i], W[i], b, sc, sh, 'tanh')
        assert float((outs[i] - ref).abs().max()) <= 2e-5


@pytest.mark.parametrize('act', ['tanh', 'relu', None])
def test_grouped_fc_every_element_over_repeated_launches(act):
    """Every output element of the grouped fused-split FC at the C2 shapes (416 tiles: most CUs run a second workgroup), over repeated
    launches, against the single-problem fp32 FC.  Guards the sporadic stale-operand events described at `epilogue_fc` (gemm_nt.hip):
    ~20 blocks of 16 rows x 1 column per launch before the fix (tools/debug/stress_fc.py is the long version)."""
    from laff_amd import ops
    torch.manual_seed(5)
    rows = [3000] * 4 + [10000] * 4
    W = [torch.randn(512, 512, device=DEV) / 22 for _ in rows]
    Ws = [ops.split_rows(w) for w in W]
    X = [torch.randn(n, 512, device=DEV) for n in rows]
    b = torch.randn(512, device=DEV) * 0.1
    sc = torch.rand(512, device=DEV) + 0.5
    sh = torch.randn(512, device=DEV) * 0.1
    probs = [dict(x=X[i], weight_split=Ws[i], bias=b, bn_scale=sc, bn_shift=sh, activation=act) for i in range(8)]
    refs = [ops.fc_act_bn(X[i], W[i], b, sc, sh, act) for i in range(8)]
    first = None
    for rep in range(12):
        outs = ops.fc_act_bn_split_grouped(probs)
        for o, r in zip(outs, refs):
            assert float((o - r).abs().max()) <= 3e-5
        if first is None:
            first = [o.clone() for o in outs]
        else:
            assert all(torch.equal(o, f) for o, f in zip(outs, first))          # launches are reproducible bit for bit


def test_fused_input_split_is_bit_identical_to_the_materialised_split():
    """laff_fc_act_bn_fused_grouped (inputs split inside the GEMM) against laff_split_rows + laff_fc_act_bn_split_grouped on the
    same big tiles: the planes are formed by the same arithmetic, so the outputs agree bit for bit wherever both paths use
    256x256 tiles -- compared here on problems without a sparse last round (no tail split on the materialised side)."""
    from laff_amd import ops
    torch.manual_seed(1)
    rows = [32768] * 8                                # 8 x 128 x 2 = 2,048 big tiles = 8 full rounds of 256 CUs
    W = [torch.randn(512, 512, device=DEV) / 22 for _ in rows]
    Ws = [ops.split_rows(w) for w in W]
    X = [torch.randn(n, 512, device=DEV) for n in rows]
    X[2][5] *= 3e4
    X[6][9] *= 1e-6
    X[3][100] = 0                                     # an all-zero row (scale 1)
    b = torch.randn(512, device=DEV) * 0.1
    sc = torch.rand(512, device=DEV) + 0.5
    sh = torch.randn(512, device=DEV) * 0.1
    probs = [dict(x=X[i], weight_split=Ws[i], bias=b, bn_scale=sc, bn_shift=sh, activation='tanh') for i in range(8)]
    fused = ops.fc_act_bn_fused_grouped(probs)
    split = ops.fc_act_bn_split_grouped(probs)
    for f, s_ in zip(fused, split):
        assert torch.equal(f, s_)
    ref = ops.fc_act_bn(X[2], W[2], b, sc, sh, 'tanh')
    assert float((fused[2] - ref).abs().max()) <= 2e-5
    # a strided view of a wider matrix and a ragged row count
    big = torch.randn(1003, 1024, device=DEV)        # 1003 rows: the row-scale pass takes rows four at a time per wavefront
    view = big[:, 256:768]
    out = ops.fc_act_bn_fused_grouped([dict(x=view, weight_split=Ws[0], bias=b, activation=None)])[0]
    ref = ops.fc_act_bn(view.contiguous(), W[0], b, None, None, None)
    assert float((out - ref).abs().max()) <= 2e-5


def test_text_row_sharding_on_one_rank_rccl_group():
    """The alternative decomposition (text-row blocks, all-gather of the video operand and of the ranks) on a 1-rank RCCL group,
    eager and with per-phase graphs: same ranks, scores and metrics as the default path."""
    import socket
    import torch.distributed as dist
    from laff_amd import synth
    from laff_amd.dist import GraphRunner, HipBackend, evaluate_sharded, evaluate_sharded_by_text
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device(DEV)
    Nt, Nv, H, d, _ = synth.WORKLOADS['c2_10kx3k']
    model = synth.build_model(H, d, dev)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev)
    backend = HipBackend(model, 'fp16')
    ref = evaluate_sharded(backend, vis, txt, gt, Nt, Nv, H)
    alone = evaluate_sharded_by_text(backend, vis, txt, gt, Nt, Nv, H)
    assert torch.equal(alone['ranks'], ref['ranks']) and torch.equal(alone['S_local'], ref['S_local'])
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, world_size=1, rank=0, device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        eager = evaluate_sharded_by_text(backend, vis, txt, gt, Nt, Nv, H, force_collectives=True)
        assert torch.equal(eager['ranks'], ref['ranks']) and torch.equal(eager['S_local'], ref['S_local'])
        np.testing.assert_allclose(eager['metrics'], ref['metrics'], rtol=1e-13)
        pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
        runner, state = GraphRunner(), {}
        for _ in range(3):
            out = evaluate_sharded_by_text(backend, vis, txt, gt, Nt, Nv, H, force_collectives=True, runner=runner, state=state,
                                           metrics_out=pinned)
            torch.cuda.synchronize()
            assert torch.equal(out['ranks'], ref['ranks']) and torch.equal(out['S_local'], ref['S_local'])
            np.testing.assert_allclose(pinned[:7].numpy(), ref['metrics'], rtol=1e-13)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------- every BASELINE config, the bench's own flags, against the ORACLE
def _oracle_sampled_parity(workload, precision, step_t, frames=False):
    """retrieval.evaluate exactly as bench.py runs it (FC_PRECISION 'fp16x3', the given similarity precision) against the numpy
    oracle on ALL videos and every step_t-th text: embeddings <= 1e-5, the sampled score rows <= 1e-4 (bf16: 2e-3), and ranks
    equal to the oracle's (float64 scores of ITS embeddings) wherever the oracle itself does not put another video within 2e-5 of
    the ground truth -- the two sets of embeddings differ by ~1e-6, nothing else may move a rank."""
    from util import oracle_towers, oracle_towers_framelaff
    import laff_amd.model.model as M
    from laff_amd import ops, retrieval, synth
    dev = torch.device(DEV)
    Nt, Nv, H, d, F = synth.WORKLOADS[workload]
    M.FC_PRECISION = 'fp16x3'
    try:
        model = synth.build_model(H, d, dev, frames=F)
        vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, frames=F)
        big = float(Nt) * Nv > 1e9
        res = retrieval.evaluate(model, vis, txt, gt, precision=precision, write_scores=not big)
    finally:
        M.FC_PRECISION = 'fp32'
    rows = np.arange(0, Nt, step_t)
    tower = oracle_towers_framelaff if frames else oracle_towers
    ve, te = tower(model, synth.to_numpy_dict(vis), synth.to_numpy_dict(txt), rows_t=rows)
    rs = torch.as_tensor(rows, device=DEV)
    assert np.abs(res.vis_emb.reshape(Nv, H, d).cpu().numpy() - ve.reshape(Nv, H, d)).max() <= 1e-5
    assert np.abs(res.txt_emb.reshape(Nt, H, d)[rs].cpu().numpy() - te.reshape(len(rows), H, d)).max() <= 1e-5
    S32 = O.txt2vis_matrix(te, ve)
    if res.S is not None:
        S_rows = res.S[rs]
    else:       # C5: the 12 GB matrix is not materialised; the sampled rows through the same GEMM
        S_rows = ops.sim_gemm(ops.pack_rows(res.txt_emb[rs].contiguous(), True, 1e-13, precision),
                              ops.pack_rows(res.vis_emb, True, 1e-13, precision), heads=H)
    assert np.abs(S_rows.cpu().numpy() - S32).max() <= (2e-3 if precision == 'bf16' else 1e-4)
    S64 = O.txt2vis_matrix_f64(te, ve)
    gs = gt.cpu().numpy()[rows]
    want = O.count_ranks(S64, gs)
    got = res.ranks[rs].cpu().numpy()
    sg = S64[np.arange(len(rows)), gs]
    nn = (np.abs(S64 - sg[:, None]) < 2e-5).sum(axis=1) - 1
    assert np.all(np.abs(got - want) <= nn), (np.abs(got - want) > nn).sum()
    assert (nn == 0).mean() > 0.3 and np.array_equal(got[nn == 0], want[nn == 0])     # (C3 is chance-level: a dense score cloud)
    # and the full rank vector against float64 scores of the HIP path's own embeddings: exact
    assert torch.equal(res.ranks[rs], _fp64_ranks(res.txt_emb.reshape(Nt, H, d), res.vis_emb.reshape(Nv, H, d), gt, rs))
    return res


def test_c2_10k_x_3k_oracle_sampled():
    _oracle_sampled_parity('c2_10kx3k', 'fp16', 50)


def test_c3_framelaff_oracle_sampled():
    _oracle_sampled_parity('c3_framelaff_10kx3k', 'fp16', 50, frames=True)


def test_c4_40k_x_10k_oracle_sampled():
    _oracle_sampled_parity('c4_40kx10k', 'fp16', 200)


def test_c5_laff_ml_oracle_sampled():
    _oracle_sampled_parity('c5_ml_100kx30k', 'bf16', 500)


def test_c5_sized_top2000_without_the_score_matrix():
    """100,000 texts x 30,000 videos x 8 heads of 512 (BASELINE config C5), the reference writer's Threshold = 2000 lists straight
    from the bf16 operands: the 12 GB score matrix is never allocated (peak extra device memory < 3 GB: 0.2 GB block buffer + the
    1.6 GB of lists + top-K scratch), and sampled rows equal the lists taken from their materialised score rows."""
    from laff_amd import ops
    Nt, Nv, H, d, K = 100000, 30000, 8, 512, 2000
    g = torch.Generator(device=DEV).manual_seed(9)
    z = torch.randn(Nv, 64, generator=g, device=DEV)
    P = torch.randn(64, H * d, generator=g, device=DEV) * 0.3
    own = torch.randint(0, Nv, (Nt,), generator=g, device=DEV)
    Ev = (z @ P + torch.randn(Nv, H * d, generator=g, device=DEV)).reshape(Nv, H, d)
    V = ops.pack_rows(Ev, True, 1e-13, 'bf16')
    del Ev
    T_parts = []
    Tbuf = torch.empty((Nt * H * d * 2,), device=DEV, dtype=torch.uint8)
    for a in range(0, Nt, 20000):
        Et = (z[own[a:a + 20000]] @ P + torch.randn(20000, H * d, generator=g, device=DEV)).reshape(20000, H, d)
        part = ops.pack_rows(Et, True, 1e-13, 'bf16')
        Tbuf[a * H * d * 2:(a + 20000) * H * d * 2] = part.buf[:20000 * H * d * 2]
        del Et, part
    T = ops.Packed(Tbuf, Nt, H * d, 'bf16', 1.0)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    idx, val = ops.topk_from_operands(T, V, K, heads=H)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < 3 * (1 << 30), peak
    assert idx.shape == (Nt, K) and bool((val[:, :-1] >= val[:, 1:]).all())
    rows = torch.arange(0, Nt, 997, device=DEV)
    for r in rows[:40].tolist():
        S = ops.sim_gemm(T.rows(r, r + 1), V, H)
        i0, v0 = ops.topk_rows(S, K)
        assert torch.equal(i0[0], idx[r]) and torch.equal(v0[0], val[r])
    # the owner video is (nearly always) the first entry
    assert float((idx[:, 0].long() == own).float().mean()) > 0.9
