"""The multi-rank orchestration (laff_amd/dist.py) on CPU: world_size 2 and 3 over gloo, with an oracle-backed
stand-in for the per-rank kernels.  Checks that sharded ranks/metrics equal the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from laff_amd.dist import evaluate_sharded, evaluate_sharded_by_text, evaluate_sharded_v16, gathered_bytes, shard_bounds
from oracle import laff_oracle as O


from dist_util import OracleBackend, problem as _problem  # noqa: E402


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        xt, xv, gt, Wt, Wv = _problem()
        t0, t1 = shard_bounds(len(xt), world, rank)
        v0, v1 = shard_bounds(len(xv), world, rank)
        res = evaluate_sharded(OracleBackend(Wt, Wv), {'x': torch.from_numpy(xv[v0:v1])}, {'x': torch.from_numpy(xt[t0:t1])},
                               torch.from_numpy(gt), len(xt), len(xv), 1)
        np.savez(os.path.join(out_dir, 'r%d.npz' % rank), ranks=res['ranks'].numpy(), metrics=np.array(res['metrics']),
                 S=res['S_local'].numpy(), col0=res['col0'])
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 10000, 40001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_equals_single(world, tmp_path):
    xt, xv, gt, Wt, Wv = _problem()
    single = evaluate_sharded(OracleBackend(Wt, Wv), {'x': torch.from_numpy(xv)}, {'x': torch.from_numpy(xt)},
                              torch.from_numpy(gt), len(xt), len(xv), 1)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cols = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), 'r%d.npz' % r))
        assert np.array_equal(z['ranks'], single['ranks'].numpy())          # uneven shards (61 / 23 rows) included
        np.testing.assert_allclose(z['metrics'], np.array(single['metrics']), rtol=0, atol=1e-12)
        cols.append(z['S'])
    np.testing.assert_allclose(np.concatenate(cols, axis=1), single['S_local'].numpy(), rtol=0, atol=1e-6)
    xt, xv, gt, Wt, Wv = _problem()
    b = OracleBackend(Wt, Wv)
    want = O.count_ranks(O.txt2vis_matrix_f64(b.embed_text({'x': torch.from_numpy(xt)}).numpy(), b.embed_video({'x': torch.from_numpy(xv)}).numpy()), gt)
    assert np.array_equal(single['ranks'].numpy(), want)


def _worker_by_text(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        xt, xv, gt, Wt, Wv = _problem()
        t0, t1 = shard_bounds(len(xt), world, rank)
        v0, v1 = shard_bounds(len(xv), world, rank)
        res = evaluate_sharded_by_text(OracleBackend(Wt, Wv), {'x': torch.from_numpy(xv[v0:v1])}, {'x': torch.from_numpy(xt[t0:t1])},
                                       torch.from_numpy(gt), len(xt), len(xv), 1)
        np.savez(os.path.join(out_dir, 't%d.npz' % rank), ranks=res['ranks'].numpy(), metrics=np.array(res['metrics']),
                 S=res['S_local'].numpy(), row0=res['row0'])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_text_row_sharding_equals_single(world, tmp_path):
    """The alternative decomposition (text-row blocks of S, all-gather of the VIDEO operand, all-gather of the ranks)."""
    xt, xv, gt, Wt, Wv = _problem()
    single = evaluate_sharded(OracleBackend(Wt, Wv), {'x': torch.from_numpy(xv)}, {'x': torch.from_numpy(xt)},
                              torch.from_numpy(gt), len(xt), len(xv), 1)
    alone = evaluate_sharded_by_text(OracleBackend(Wt, Wv), {'x': torch.from_numpy(xv)}, {'x': torch.from_numpy(xt)},
                                     torch.from_numpy(gt), len(xt), len(xv), 1)
    assert np.array_equal(alone['ranks'].numpy(), single['ranks'].numpy())
    mp.spawn(_worker_by_text, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rows = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), 't%d.npz' % r))
        assert np.array_equal(z['ranks'], single['ranks'].numpy())          # uneven shards (61 / 23 rows) included
        np.testing.assert_allclose(z['metrics'], np.array(single['metrics']), rtol=0, atol=1e-12)
        rows.append(z['S'])
    np.testing.assert_allclose(np.concatenate(rows, axis=0), single['S_local'].numpy(), rtol=0, atol=1e-6)


def _worker_v16(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        xt, xv, gt, Wt, Wv = _problem()
        t0, t1 = shard_bounds(len(xt), world, rank)
        v0, v1 = shard_bounds(len(xv), world, rank)
        res = evaluate_sharded_v16(OracleBackend(Wt, Wv), {'x': torch.from_numpy(xv[v0:v1])}, {'x': torch.from_numpy(xt[t0:t1])},
                                   torch.from_numpy(gt), len(xt), len(xv), 1, pair_bucket_cap=64)
        np.savez(os.path.join(out_dir, 'h%d.npz' % rank), ranks=res['ranks'].numpy(), metrics=np.array(res['metrics']),
                 S=res['S_local'].numpy(), col0=res['col0'], fill=res['pair_fill'].numpy(),
                 bytes=np.array(sorted(res['gathered_bytes'].items()), dtype=object), allow_pickle=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_video16_equals_single(world, tmp_path):
    """'video16': the 16-bit text operand + the fp32 video rows are gathered, the in-band pairs travel to the owner of their text
    row (all-to-all of 8-byte pairs) and are re-scored there -- ranks equal to the single-process ones, uneven shards (61 / 23 rows),
    and pairs really did travel."""
    xt, xv, gt, Wt, Wv = _problem()
    single = evaluate_sharded(OracleBackend(Wt, Wv), {'x': torch.from_numpy(xv)}, {'x': torch.from_numpy(xt)},
                              torch.from_numpy(gt), len(xt), len(xv), 1)
    alone = evaluate_sharded_v16(OracleBackend(Wt, Wv), {'x': torch.from_numpy(xv)}, {'x': torch.from_numpy(xt)},
                                 torch.from_numpy(gt), len(xt), len(xv), 1, pair_bucket_cap=4096)
    assert np.array_equal(alone['ranks'].numpy(), single['ranks'].numpy())
    assert int(alone['pair_fill'][0]) > 0                                   # the band is not empty: the resolve path is exercised
    mp.spawn(_worker_v16, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cols, moved = [], 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), 'h%d.npz' % r), allow_pickle=True)
        assert np.array_equal(z['ranks'], single['ranks'].numpy())
        np.testing.assert_allclose(z['metrics'], np.array(single['metrics']), rtol=0, atol=1e-12)
        cols.append(z['S'])
        assert z['fill'][world] == 0                                        # no bucket overflowed
        moved += int(z['fill'][:world].sum()) - int(z['fill'][r])          # pairs whose text row lives on another rank
    assert moved > 0
    np.testing.assert_allclose(np.concatenate(cols, axis=1), single['S_local'].numpy(), rtol=0, atol=1e-6)
    b = gathered_bytes('video16', 40000, 10000, 512, 8, 10240)
    assert b['all_gather_text_16bit'] * 2 == gathered_bytes('video', 40000, 10000, 512, 8)['all_gather_text_fp32']


def test_video16_full_bucket_poisons_the_ranks():
    """a pair bucket too small for the band: flagged, and rank 0 of the result is < 1 (what the metrics kernel turns into an error)"""
    xt, xv, gt, Wt, Wv = _problem()
    res = evaluate_sharded_v16(OracleBackend(Wt, Wv), {'x': torch.from_numpy(xv)}, {'x': torch.from_numpy(xt)},
                               torch.from_numpy(gt), len(xt), len(xv), 1, pair_bucket_cap=4, want_metrics=False)
    assert int(res['pair_fill'][1]) == 1 and int(res['ranks'][0]) < 1


@pytest.mark.parametrize('world,rank', [(3, 0), (3, 2), (8, 5)])
def test_emulated_rank_of_a_sharded_pass(world, rank):
    """EmulatedComm (bench.py --emulate-shard): one rank's share of a `world`-rank pass in a single process, the collectives replaced by
    what they would leave on this rank.  With the peers' contributions pre-filled the emulated rank reproduces the single-process
    ranks in both decompositions (uneven shards: 61 texts / 23 videos)."""
    from laff_amd.dist import EmulatedComm
    xt, xv, gt, Wt, Wv = _problem()
    Nt, Nv = len(xt), len(xv)
    be = OracleBackend(Wt, Wv)
    gt_t = torch.from_numpy(gt)
    single = evaluate_sharded(be, {'x': torch.from_numpy(xv)}, {'x': torch.from_numpy(xt)}, gt_t, Nt, Nv, 1)
    t0, t1 = shard_bounds(Nt, world, rank)
    v0, v1 = shard_bounds(Nv, world, rank)
    vis_l, txt_l = {'x': torch.from_numpy(xv[v0:v1])}, {'x': torch.from_numpy(xt[t0:t1])}
    tmax = max(b - a for a, b in (shard_bounds(Nt, world, r) for r in range(world)))
    vmax = max(b - a for a, b in (shard_bounds(Nv, world, r) for r in range(world)))

    def padded(E, n, nmax):
        out = torch.zeros((world * nmax, E.shape[1]), dtype=E.dtype)
        for r in range(world):
            a, b = shard_bounds(n, world, r)
            out[r * nmax:r * nmax + (b - a)] = E[a:b]
        return out

    # 'video': the text rows of the peers are in the gather buffer, the peers' exact ground-truth scores and counts in `peers`
    full_prep = be.prepare(single['txt_emb'], single['vis_emb'], None, None, gt_t, 0)
    state = {'gathered': padded(single['txt_emb'].reshape(Nt, -1), Nt, tmax)}
    state['gathered'][rank * tmax:(rank + 1) * tmax] = -7.0                      # this rank's slot is filled by the pass itself
    comm = EmulatedComm(world, rank, {'max': full_prep.s_gt64.clone()})
    first = evaluate_sharded(be, vis_l, txt_l, gt_t, Nt, Nv, 1, state=state, comm_impl=comm, want_metrics=False)
    comm.peers['sum'] = (single['ranks'] - first['ranks']).to(torch.int32)      # what the other ranks' blocks add
    res = evaluate_sharded(be, vis_l, txt_l, gt_t, Nt, Nv, 1, state=state, comm_impl=comm)
    assert np.array_equal(res['ranks'].numpy(), single['ranks'].numpy())
    np.testing.assert_allclose(res['metrics'], np.array(single['metrics']), rtol=0, atol=1e-12)
    np.testing.assert_allclose(res['S_local'].numpy(), single['S_local'].numpy()[:, v0:v1], rtol=0, atol=1e-6)
    assert int((first['ranks'] <= single['ranks']).all())

    # 'text': the video rows of the peers are in the gather buffer, the peers' ranks in the rank buffer
    state = {'gathered_v': padded(single['vis_emb'].reshape(Nv, -1), Nv, vmax),
             'gathered_r': padded(single['ranks'].to(torch.int32)[:, None], Nt, tmax).reshape(-1)}
    state['gathered_v'][rank * vmax:(rank + 1) * vmax] = -7.0
    state['gathered_r'][rank * tmax:(rank + 1) * tmax] = 0
    res = evaluate_sharded_by_text(be, vis_l, txt_l, gt_t, Nt, Nv, 1, state=state, comm_impl=EmulatedComm(world, rank))
    assert np.array_equal(res['ranks'].numpy(), single['ranks'].numpy())
    np.testing.assert_allclose(res['S_local'].numpy(), single['S_local'].numpy()[t0:t1], rtol=0, atol=1e-6)


def test_collectives_refuse_an_open_capture(monkeypatch):
    """TorchComm's collectives are eager calls between the captured phases; issued while a HIP-graph capture is open they raise instead
    of being recorded (replayed without their peers).  The capture state is stubbed: there is no GPU in the CPU suite."""
    from laff_amd.dist import TorchComm

    class FakeCuda:
        is_cuda = True

        def __init__(self, n):
            self.shape = (n,)
    monkeypatch.setattr(torch.cuda, 'is_current_stream_capturing', lambda: True)
    c = TorchComm()
    for call in (lambda: c.all_gather(FakeCuda(8), FakeCuda(4)), lambda: c.all_reduce(FakeCuda(4), 'sum'),
                 lambda: c.all_to_all(FakeCuda(4), FakeCuda(4))):
        with pytest.raises(RuntimeError, match='capture is open'):
            call()
    # CPU tensors (the gloo tests) are never inside a capture
    TorchComm._not_capturing('all_reduce', torch.zeros(2))
