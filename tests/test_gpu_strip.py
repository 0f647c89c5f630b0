"""The strip form of the K = 512 similarity GEMM (laff_amd/csrc/sim_strip.hip: one wavefront per SIMD holding 64 text rows in
registers, video blocks streamed through LDS) against the tiled kernel, the oracle and float64 ranks.  Both kernels sit behind the
same entry points (laff_sim_gemm / laff_sim_gemm_banded, include/laff_hip.h); LAFF_STRIP picks: 0 = tiled only, 1 = strip where it
is faster (default), 2 = also bf16 operands and smaller problems, 3 = wherever it can run (score rows of any pitch)."""
import os

import numpy as np
import pytest
import torch

from oracle import laff_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture
def strip_mode():
    from laff_amd import ops

    def set_mode(m):
        os.environ['LAFF_STRIP'] = str(m)
        ops.reset_contexts()
    yield set_mode
    os.environ.pop('LAFF_STRIP', None)
    ops.reset_contexts()


def _embeddings(Nt, Nv, noise, seed, H=1, d=512):
    g = torch.Generator(device=DEV).manual_seed(seed)
    z = torch.randn(Nv, 48, generator=g, device=DEV)
    P = torch.randn(48, H * d, generator=g, device=DEV)
    gt = (torch.arange(Nt, device=DEV) * 7919 % Nv).to(torch.int32)
    Ev = (z @ P + noise * torch.randn(Nv, H * d, generator=g, device=DEV)).reshape(Nv, H, d).contiguous()
    Et = (z[gt.long()] @ P + noise * torch.randn(Nt, H * d, generator=g, device=DEV)).reshape(Nt, H, d).contiguous()
    return Et, Ev, gt


def _fp64_count(Et, Ev, gt):
    t, v = Et.double(), Ev.double()
    t = t / (t.pow(2).sum(-1, keepdim=True).sqrt() + (1e-13 + 1e-14))
    v = v / (v.pow(2).sum(-1, keepdim=True).sqrt() + (1e-13 + 1e-14))
    out = torch.empty(Et.shape[0], dtype=torch.int32, device=Et.device)
    for a in range(0, Et.shape[0], 4096):
        S = torch.einsum('thd,vhd->tv', t[a:a + 4096], v) / Et.shape[1]
        g = gt[a:a + 4096].long()
        ab = S > S.gather(1, g[:, None])
        ab[torch.arange(ab.shape[0], device=S.device), g] = False
        out[a:a + 4096] = ab.sum(1).to(torch.int32)
    return out


def _run(ops, Et, Ev, gt, prec, want_scores, ldo=None, pair_cap=None, prescale=None):
    T, V = ops.pack_rows(Et, True, 1e-13, prec, prescale), ops.pack_rows(Ev, True, 1e-13, prec, prescale)
    st = ops.rank_prepare(Et, Ev, T, V, gt, 0, pair_cap)
    out = None
    if want_scores and ldo is not None:
        out = torch.full((Et.shape[0], ldo), -7.0, device=DEV)[:, :Ev.shape[0]]
    S = ops.sim_gemm_banded(st, want_scores, out=out)
    ops.rank_resolve(st, S)
    torch.cuda.synchronize()
    return S, st


@pytest.mark.parametrize('prec,prescale', [('fp16', None), ('fp16', 64.0), ('bf16', None)])
def test_strip_equals_tiled_at_c4(strip_mode, prec, prescale):
    """40000 x 10000 (BASELINE config C4): scores bit-identical to the tiled kernel's, same band pairs, counts equal to the float64
    ranks; the default (no prescale) is the scale == 1 instantiation (no multiply in the epilogue), prescale 64 the general one."""
    from laff_amd import ops
    Et, Ev, gt = _embeddings(40000, 10000, 9.0, 3)
    want = _fp64_count(Et, Ev, gt)
    strip_mode(0)
    S0, st0 = _run(ops, Et, Ev, gt, prec, True, prescale=prescale)
    p0, c0 = st0.pair_indices(), st0.count.clone()
    assert torch.equal(c0, want)
    strip_mode(2)
    S1, st1 = _run(ops, Et, Ev, gt, prec, True, prescale=prescale)
    assert not st1.listed_pairs()[1]
    assert int(st1._header()[2]) >> 31 == 1, 'the strip kernel did not run'
    assert torch.equal(S1, S0)
    assert torch.equal(st1.count, want)
    assert set(map(tuple, st1.pair_indices().tolist())) == set(map(tuple, p0.tolist()))
    _, st2 = _run(ops, Et, Ev, gt, prec, False, prescale=prescale)
    assert int(st2._header()[2]) >> 31 == 1
    assert torch.equal(st2.count, want)
    # scores alone (laff_sim_gemm)
    T, V = ops.pack_rows(Et, True, 1e-13, prec, prescale), ops.pack_rows(Ev, True, 1e-13, prec, prescale)
    Sp1 = ops.sim_gemm(T, V)
    strip_mode(0)
    assert torch.equal(ops.sim_gemm(T, V), Sp1)


@pytest.mark.parametrize('Nt,Nv,ldo', [(2309, 8200, 8200), (2305, 8197, 8200), (4099, 4133, 4160), (2049, 8193, 8196)])
def test_strip_ragged_shapes_vs_oracle(strip_mode, Nt, Nv, ldo):
    """Partial last strip, partial last column block, padded score rows (the padding must stay untouched), checked against the oracle:
    scores within the fp16 operand bound of oracle/laff_oracle.txt2vis_matrix_f64 (model/model.py:1003-1016), counts exactly its ranks."""
    from laff_amd import ops
    Et, Ev, gt = _embeddings(Nt, Nv, 6.0, Nt + Nv)
    S64 = O.txt2vis_matrix_f64(Et.cpu().numpy(), Ev.cpu().numpy())
    want = O.count_ranks(S64, gt.cpu().numpy()) - 1
    strip_mode(3)
    S1, st1 = _run(ops, Et, Ev, gt, 'fp16', True, ldo=ldo)
    assert int(st1._header()[2]) >> 31 == 1, 'the strip kernel did not run'
    assert not st1.listed_pairs()[1]
    np.testing.assert_array_equal(st1.count.cpu().numpy(), want)
    assert float(np.abs(S1.cpu().numpy().astype(np.float64) - S64).max()) < 5e-4          # include/laff_hip.h: LAFF_PREC_FP16
    if ldo > Nv:
        assert bool((S1._base[:, Nv:] == -7.0).all()) if S1._base is not None else True
    strip_mode(0)
    S0, st0 = _run(ops, Et, Ev, gt, 'fp16', True, ldo=ldo)
    assert int(st0._header()[2]) >> 31 == 0
    assert torch.equal(S0, S1)
    strip_mode(3)
    _, st2 = _run(ops, Et, Ev, gt, 'fp16', False)
    np.testing.assert_array_equal(st2.count.cpu().numpy(), want)


def test_strip_list_overflow_is_flagged_and_poisons(strip_mode):
    """A dump list too small for the band: the flag is raised, count[0] is poisoned (include/laff_hip.h) and rank_metrics refuses."""
    from laff_amd import ops
    Et, Ev, gt = _embeddings(8192, 8192, 0.02, 11)           # near-duplicate rows: many scores inside the band
    strip_mode(2)
    _, st = _run(ops, Et, Ev, gt, 'fp16', False, pair_cap=4 * (1 + 64 * 24) * 40)
    assert int(st._header()[2]) >> 31 == 1
    n, over = st.listed_pairs()
    if not over:
        pytest.skip('list did not overflow with this data (%d pairs)' % n)
    assert int(st.count[0]) < -(1 << 25)
    with pytest.raises(Exception):
        ops.rank_metrics(st.count, base=1)
    # the same data with room: exact
    _, st_ok = _run(ops, Et, Ev, gt, 'fp16', False, pair_cap=64 << 20)
    assert not st_ok.listed_pairs()[1]
    assert torch.equal(st_ok.count, _fp64_count(Et, Ev, gt))


def test_default_dispatch_takes_the_strip_kernel_at_c4_and_not_for_small_or_bf16(strip_mode):
    from laff_amd import ops
    strip_mode(1)
    Et, Ev, gt = _embeddings(40000, 10000, 9.0, 5)
    _, st = _run(ops, Et, Ev, gt, 'fp16', False)
    assert int(st._header()[2]) >> 31 == 1
    _, st = _run(ops, Et, Ev, gt, 'bf16', False)
    assert int(st._header()[2]) >> 31 == 0
    Et, Ev, gt = _embeddings(3000, 2000, 9.0, 5)
    _, st = _run(ops, Et, Ev, gt, 'fp16', False)
    assert int(st._header()[2]) >> 31 == 0
    assert torch.equal(st.count, _fp64_count(Et, Ev, gt))


def _fp64_v2t_count(Et, Ev, owner):
    """count[t] = #{t' != t : S64[t', v] > S64[t, v]}, v = owner[t], on the float64 cosine scores (torch, column blocks)."""
    t, v = Et.double(), Ev.double()
    t = t / (t.pow(2).sum(-1, keepdim=True).sqrt() + (1e-13 + 1e-14))
    v = v / (v.pow(2).sum(-1, keepdim=True).sqrt() + (1e-13 + 1e-14))
    Nt = Et.shape[0]
    out = torch.empty(Nt, dtype=torch.int32, device=Et.device)
    ow = owner.long()
    for a in range(0, Ev.shape[0], 512):
        S = torch.einsum('thd,vhd->tv', t, v[a:a + 512]) / Et.shape[1]          # (Nt, 512)
        mine = ((ow >= a) & (ow < a + 512)).nonzero().flatten()
        if mine.numel() == 0:
            continue
        col = (ow[mine] - a)
        thr = S[mine, col]                                                       # (n,)
        for b in range(0, mine.numel(), 2048):
            m, c, th = mine[b:b + 2048], col[b:b + 2048], thr[b:b + 2048]
            ab = S[:, c] > th[None, :]                                           # (Nt, n)
            ab[m, torch.arange(m.numel(), device=S.device)] = False
            out[m] = ab.sum(0).to(torch.int32)
    return out


@pytest.mark.parametrize('prec', ['fp16', 'bf16'])
def test_exact_v2t_positions_at_c1_shapes(strip_mode, prec):
    """59800 captions x 2990 videos, 20 captions per video (the reference's MSR-VTT test split, BASELINE config C1): the video->text
    counts of laff_v2t_count_exact equal those of the float64 scores with fp16 and with bf16 operands."""
    from laff_amd import ops
    Nv, per = 2990, 20
    Nt = Nv * per
    g = torch.Generator(device=DEV).manual_seed(21)
    z = torch.randn(Nv, 48, generator=g, device=DEV)
    P = torch.randn(48, 512, generator=g, device=DEV)
    owner = (torch.arange(Nt, device=DEV) // per).to(torch.int32)
    Ev = (z @ P + 7.0 * torch.randn(Nv, 512, generator=g, device=DEV)).reshape(Nv, 1, 512).contiguous()
    Et = (z[owner.long()] @ P + 7.0 * torch.randn(Nt, 512, generator=g, device=DEV)).reshape(Nt, 1, 512).contiguous()
    want = _fp64_v2t_count(Et, Ev, owner)
    strip_mode(1)
    T, V = ops.pack_rows(Et, True, 1e-13, prec), ops.pack_rows(Ev, True, 1e-13, prec)
    S, count, st = ops.exact_ranks(Et, Ev, T, V, owner)
    assert not st.listed_pairs()[1]
    off = torch.arange(0, Nt + 1, per, device=DEV, dtype=torch.int32)
    idx = torch.arange(Nt, device=DEV, dtype=torch.int32)
    got = ops.v2t_count_exact(S, st, off, idx, per)
    assert torch.equal(got, want)
    assert torch.equal(ops.v2t_count_exact(S, st, off, idx, per, list_cap=64), want)          # overflow -> one retry with the wanted size
    assert torch.equal(count, _fp64_count(Et, Ev, owner))


@pytest.mark.parametrize('Nv,sizes,prec', [(97, (1,), 'fp16'), (130, (0, 3, 1), 'fp16'), (64, (7, 2, 8, 0, 5), 'bf16'), (40, (20, 0, 13), 'fp16x3'),
                                           (33, (4,), 'fp32')])
def test_exact_v2t_ragged_groups_vs_oracle(Nv, sizes, prec):
    """Videos with 0 .. 20 captions in every mix (group passes of 4, 8 and 16 thresholds; columns without a caption), near-duplicate
    embeddings so that many scores sit inside the band: counts equal those of oracle.txt2vis_matrix_f64 column by column."""
    from laff_amd import ops
    g = np.random.default_rng(Nv)
    per = np.array([sizes[v % len(sizes)] for v in range(Nv)])
    owner = np.repeat(np.arange(Nv), per).astype(np.int32)
    perm = g.permutation(owner.size)
    owner = owner[perm]                                            # captions of a video are not consecutive
    Nt, H, d = owner.size, 2, 64
    base = g.normal(0, 1, (1, H, d))
    ev = (base + 0.05 * g.normal(0, 1, (Nv, H, d))).astype(np.float32)
    et = (ev[owner] + 0.03 * g.normal(0, 1, (Nt, H, d))).astype(np.float32)
    S64 = O.txt2vis_matrix_f64(et, ev)
    want = np.array([int(np.sum(S64[:, owner[t]] > S64[t, owner[t]])) for t in range(Nt)])
    Et, Ev = torch.as_tensor(et, device=DEV), torch.as_tensor(ev, device=DEV)
    gt = torch.as_tensor(owner, device=DEV)
    T, V = ops.pack_rows(Et, True, 1e-13, prec), ops.pack_rows(Ev, True, 1e-13, prec)
    S, count, st = ops.exact_ranks(Et, Ev, T, V, gt)
    order = np.argsort(owner, kind='stable').astype(np.int32)
    off = np.zeros(Nv + 1, np.int32)
    np.cumsum(np.bincount(owner, minlength=Nv), out=off[1:])
    got = ops.v2t_count_exact(S, st, torch.as_tensor(off, device=DEV), torch.as_tensor(order, device=DEV), int(per.max()))
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    np.testing.assert_array_equal(count.cpu().numpy(), O.count_ranks(S64, owner) - 1)


def test_misaligned_score_rows_go_to_the_tiled_kernel(strip_mode):
    """A caller's score matrix whose row pitch is not a multiple of 64 bytes stays on the tiled kernel in the default mode (the strip
    kernel's 128-byte row pieces would straddle lines at odd offsets: 2 x slower); same scores either way."""
    from laff_amd import ops
    Et, Ev, gt = _embeddings(40000, 10000, 9.0, 8)
    strip_mode(1)
    S_a, st_a = _run(ops, Et, Ev, gt, 'fp16', True, ldo=10008)
    assert int(st_a._header()[2]) >> 31 == 0
    S_b, st_b = _run(ops, Et, Ev, gt, 'fp16', True, ldo=10016)
    assert int(st_b._header()[2]) >> 31 == 1
    assert torch.equal(S_a, S_b) and torch.equal(st_a.count, st_b.count)


def test_strip_kernel_on_video_shards(strip_mode):
    """The 'video' decomposition of laff_amd/dist.py on one GPU, strip kernel: every shard counts against the all-reduced (MAX) exact
    ground-truth scores, most ground-truth columns lie outside the shard (col0 != 0); the summed counts are the float64 ranks."""
    from laff_amd import ops
    Et, Ev, gt = _embeddings(40000, 10000, 9.0, 12)
    want = _fp64_count(Et, Ev, gt)
    strip_mode(2)
    T = ops.pack_rows(Et, True, 1e-13, 'fp16')
    bounds = [0, 3333, 3334, 10000]
    states = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        Evs = Ev[a:b].contiguous()
        states.append(ops.rank_prepare(Et, Evs, T, ops.pack_rows(Evs, True, 1e-13, 'fp16'), gt, col0=a))
    s_all = torch.stack([s.s_gt64 for s in states]).max(dim=0).values
    total = torch.zeros(40000, dtype=torch.int32, device=DEV)
    used = []
    for s in states:
        s.s_gt64.copy_(s_all)
        S = ops.sim_gemm_banded(s, True)
        used.append(int(s._header()[2]) >> 31)
        total += ops.rank_resolve(s, S)
        assert not s.listed_pairs()[1]
    assert used == [1, 0, 1]                     # (the one-column shard is far below the strip kernel's size)
    assert torch.equal(total, want)


@pytest.mark.parametrize('prec,H,d', [('fp16', 1, 512), ('bf16', 1, 512), ('fp16', 8, 128), ('bf16', 2, 512), ('fp16', 3, 260)])
def test_fused_prepare_leaves_what_rank_prepare_leaves(prec, H, d):
    """laff_fuse_packed_rank: the two fuse launches of a pass (videos first) do laff_rank_prepare's work -- s_gt64 BIT-equal (same
    arithmetic as the re-score, also with the heads of a row coming from different workgroups), bands equal to rounding (never
    narrower than 1 - 1e-6 of rank_prepare's), count / list header cleared -- and the pipeline that follows gives the float64 ranks."""
    from laff_amd import ops
    Nt, Nv, L = 6001, 2103, 3
    g = torch.Generator(device=DEV).manual_seed(5)
    z = torch.randn(Nv, 32, generator=g, device=DEV)
    gt = (torch.arange(Nt, device=DEV) * 31 % Nv).to(torch.int32)
    w = torch.randn(H, d, generator=g, device=DEV) * 0.05
    b, gw = torch.zeros(H, device=DEV), torch.ones(H, device=DEV)

    def planes(n, lat):
        return [(lat @ torch.randn(32, H * d, generator=g, device=DEV) + 3.0 * torch.randn(n, H * d, generator=g, device=DEV), False, None, None)
                for _ in range(L)]
    pv, pt = planes(Nv, z), planes(Nt, z[gt.long()])
    flags = ops.attention_flags(True, False)
    fp = ops.FusedPrepare(Nt, Nv, gt, heads=H)
    fp.count.fill_(7)
    fp.pairs[:4] = 9
    Ev, V = ops.fuse(pv, H, d, w, b, gw, flags, packed_precision=prec, rank_side=fp.video)
    Et, T = ops.fuse(pt, H, d, w, b, gw, flags, packed_precision=prec, rank_side=fp.text)
    st = fp.state()
    ref = ops.rank_prepare(Et, Ev, T, V, gt)
    assert torch.equal(st.s_gt64, ref.s_gt64)
    assert int(st.count.abs().sum()) == 0 and st.pairs[:4].tolist() == [0, 0, 0, 0]
    nb = ((Nv + 3) & ~3) + (Nv + 63) // 64
    for a, r in ((st.band_t[:Nt], ref.band_t[:Nt]), (st.band_v[:Nv], ref.band_v[:Nv]), (st.band_v[(Nv + 3) & ~3:nb], ref.band_v[(Nv + 3) & ~3:nb])):
        assert float(((a - r).abs() / r).max()) < 1e-5 and bool((a >= r * (1 - 1e-6)).all())
    S = ops.sim_gemm_banded(st, True)
    ops.rank_resolve(st, S)
    assert not st.listed_pairs()[1]
    assert torch.equal(st.count, _fp64_count(Et, Ev, gt))
    # a text whose video is not among these columns: -inf, like rank_prepare
    fp2 = ops.FusedPrepare(Nt, Nv, gt, col0=100, heads=H)
    ops.fuse(pv, H, d, w, b, gw, flags, packed_precision=prec, rank_side=fp2.video)
    ops.fuse(pt, H, d, w, b, gw, flags, packed_precision=prec, rank_side=fp2.text)
    out = (gt < 100) | (gt >= 100 + Nv)
    assert bool(torch.isneginf(fp2.s_gt64[out]).all()) and bool(torch.isfinite(fp2.s_gt64[~out]).all())
