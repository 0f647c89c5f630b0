"""'video16' (laff_amd.dist.evaluate_sharded_v16) on the GPU: the HIP kernels behind it (laff_rank_prepare_part, laff_rank_export_pairs,
laff_rank_resolve on exported buckets) against the single-GPU exact ranks -- whole pass on one rank, every rank's part of a
3-rank pass emulated on one device (uneven shards, both list formats of the banded GEMM), and the N > 1 code path over a 1-rank RCCL
group (eager + per-phase HIP graphs).  The loop being sharded is /root/reference/model/model.py:1057-1077 + predictor.py:232-244."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _problem(name, seed=0):
    from laff_amd import synth
    Nt, Nv, H, d, _ = synth.WORKLOADS[name]
    dev = torch.device(DEV)
    model = synth.build_model(H, d, dev)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev)
    return model, vis, txt, gt, Nt, Nv, H


@pytest.mark.parametrize('name,prec', [('tiny', 'fp16'), ('c2_10kx3k', 'fp16'), ('c2_10kx3k', 'bf16')])
def test_video16_single_rank_equals_the_plain_pass(name, prec):
    from laff_amd.dist import HipBackend, evaluate_sharded, evaluate_sharded_v16
    model, vis, txt, gt, Nt, Nv, H = _problem(name)
    backend = HipBackend(model, prec)
    ref = evaluate_sharded(backend, vis, txt, gt, Nt, Nv, H)
    got = evaluate_sharded_v16(backend, vis, txt, gt, Nt, Nv, H)
    assert torch.equal(got['ranks'], ref['ranks'])
    np.testing.assert_allclose(got['metrics'], ref['metrics'], rtol=1e-13)
    assert int(got['pair_fill'][0]) > 0 and int(got['pair_fill'][1]) == 0
    # (the listed pairs keep the GEMM's own value here: the exact score is computed by the owner of the text row)
    assert float((got['S_local'] - ref['S_local']).abs().max()) <= (2e-4 if prec == 'fp16' else 3e-3)


@pytest.mark.parametrize('Nt,Nv,prec,world', [(3001, 701, 'fp16', 3), (40000, 10000, 'fp16', 3), (10000, 3000, 'bf16', 2)])
def test_video16_every_rank_of_a_sharded_pass_emulated_on_one_device(Nt, Nv, prec, world):
    """What each of `world` ranks would run, one after the other on this device, the collectives replaced by slicing / concatenation:
    ranks torch.equal to the single-GPU exact ranks.  40000 x 10000 sends the video shards' GEMMs through the strip kernel (dumped
    groups), the others through the tiled kernel's plain list; shards are uneven."""
    from laff_amd import ops, retrieval, synth
    from laff_amd.dist import shard_bounds
    dev = torch.device(DEV)
    model = synth.build_model(1, 512, dev, seed=3)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, seed=3)
    with torch.no_grad():
        Ev, Et = retrieval.embed(model, vis, txt)
    Et, Ev = Et.reshape(Nt, 1, -1).contiguous(), Ev.reshape(Nv, 1, -1).contiguous()
    T_all = ops.pack_rows(Et, True, 1e-13, prec)
    V_all = ops.pack_rows(Ev, True, 1e-13, prec)
    _, ref_count, _ = ops.exact_ranks(Et, Ev, T_all, V_all, gt, want_scores=False)
    tb = [shard_bounds(Nt, world, r) for r in range(world)]
    vb = [shard_bounds(Nv, world, r) for r in range(world)]
    bounds = torch.tensor([lo for lo, _ in tb] + [Nt], dtype=torch.int32, device=dev)
    # text owners: exact ground-truth scores and bands of their rows (every fp32 video row is there)
    s_parts, b_parts = [], []
    for t0, t1 in tb:
        s, b = ops.rank_prepare_text(Et[t0:t1].contiguous(), Ev, T_all.rows(t0, t1), gt[t0:t1].contiguous(), 0)
        s_parts.append(s)
        b_parts.append(b[:t1 - t0])
    s_all = torch.cat(s_parts)
    b_all = torch.zeros(Nt + 4, device=dev)
    b_all[:Nt] = torch.cat(b_parts)
    # video owners: banded GEMM of all texts x their videos, export
    cap = 1 << 15
    outs, fills, total = [], [], torch.zeros(Nt, dtype=torch.int32, device=dev)
    for v0, v1 in vb:
        Vr = V_all.rows(v0, v1)
        band_v = ops.rank_band_video(Ev[v0:v1].contiguous(), Vr)
        st = ops.banded_state(T_all, Vr, 1, gt, v0, s_all, b_all, band_v)
        ops.sim_gemm_banded(st, False)
        out, fill = ops.rank_export_pairs(st, None, bounds, v0, cap)
        assert int(fill[world]) == 0
        outs.append(out)
        fills.append(fill)
        total += st.count
    assert sum(int(f[:world].sum()) for f in fills) > 0
    # text owners: the pairs of their rows from every video owner, exact re-score
    got = torch.empty_like(total)
    for o, (t0, t1) in enumerate(tb):
        lst = torch.cat([torch.tensor([0, 0, world * cap, 4], dtype=torch.int32, device=dev)] + [out[o].reshape(-1) for out in outs])
        mine = total[t0:t1].clone()
        ops.rank_resolve_list(Et[t0:t1].contiguous(), Ev, s_parts[o], mine, lst)
        got[t0:t1] = mine
    assert torch.equal(got, ref_count)


def test_video16_under_a_graph_runner_without_collectives():
    """world == 1, no process group, per-phase HIP graphs: the resolve phase is captured once and must find this step's pair list at
    the address it captured (the list lives in `state` and is refilled inside the phase).  Allocations between the replays move the
    caching allocator's blocks around: a list allocated per call would be read after free."""
    from laff_amd.dist import GraphRunner, HipBackend, evaluate_sharded, evaluate_sharded_v16
    model, vis, txt, gt, Nt, Nv, H = _problem('c2_10kx3k')
    backend = HipBackend(model, 'fp16')
    ref = evaluate_sharded(backend, vis, txt, gt, Nt, Nv, H)
    pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
    runner, state = GraphRunner(), {}
    junk = []
    for k in range(4):                           # capture, then three replays
        out = evaluate_sharded_v16(backend, vis, txt, gt, Nt, Nv, H, runner=runner, state=state, metrics_out=pinned)
        torch.cuda.synchronize()
        assert torch.equal(out['ranks'], ref['ranks'])
        np.testing.assert_allclose(pinned[:7].numpy(), ref['metrics'], rtol=1e-13)
        assert pinned[7].item() == 0
        junk.append(torch.full((1 << 18,), -1, dtype=torch.int32, device=DEV))      # lands where a freed per-call list would have been
        junk.append(torch.full((4 + 2 * 4096 * (k + 1),), -1, dtype=torch.int32, device=DEV))


def test_video16_refuses_split_operands():
    from laff_amd.dist import HipBackend, evaluate_sharded_v16
    model, vis, txt, gt, Nt, Nv, H = _problem('tiny')
    with pytest.raises(ValueError, match='16-bit'):
        evaluate_sharded_v16(HipBackend(model, 'fp16x3'), vis, txt, gt, Nt, Nv, H)


def test_video16_on_one_rank_rccl_group():
    """The N > 1 code path of 'video16' (three all-gathers, the pair all-to-all, the count all-reduce, per-phase HIP graphs) on a 1-rank
    RCCL group: same ranks and metrics as the plain single-GPU pass."""
    import socket
    import torch.distributed as dist
    from laff_amd.dist import GraphRunner, HipBackend, evaluate_sharded, evaluate_sharded_v16
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    model, vis, txt, gt, Nt, Nv, H = _problem('c2_10kx3k')
    backend = HipBackend(model, 'fp16')
    ref = evaluate_sharded(backend, vis, txt, gt, Nt, Nv, H)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, world_size=1, rank=0, device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        eager = evaluate_sharded_v16(backend, vis, txt, gt, Nt, Nv, H, force_collectives=True)
        assert torch.equal(eager['ranks'], ref['ranks'])
        np.testing.assert_allclose(eager['metrics'], ref['metrics'], rtol=1e-13)
        pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
        runner, state = GraphRunner(), {}
        for _ in range(3):                       # capture, then two replays
            out = evaluate_sharded_v16(backend, vis, txt, gt, Nt, Nv, H, force_collectives=True, runner=runner, state=state, metrics_out=pinned)
            torch.cuda.synchronize()
            assert torch.equal(out['ranks'], ref['ranks'])
            np.testing.assert_allclose(pinned[:7].numpy(), ref['metrics'], rtol=1e-13)
            assert pinned[7].item() == 0
    finally:
        dist.destroy_process_group()
