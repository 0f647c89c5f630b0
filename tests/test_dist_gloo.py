"""The multi-rank orchestration (laff_amd/dist.py) on CPU: world_size 2 and 3 over gloo, with an oracle-backed
stand-in for the per-rank kernels.  Checks that sharded ranks/metrics equal the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from laff_amd.dist import evaluate_sharded, evaluate_sharded_by_text, shard_bounds
from oracle import laff_oracle as O


class Packed:
    def __init__(self, buf, N, K):
        self.buf, self.N, self.K, self.precision, self.prescale = buf, N, K, 'fp16', 1.0


class OracleBackend:
    """CPU stand-in with the HipBackend interface: fp16-rounded operands, fp32 GEMM."""

    def __init__(self, Wt, Wv):
        self.Wt, self.Wv = Wt, Wv

    def embed_text(self, f):
        return torch.from_numpy(O.l2norm(np.tanh(f['x'].numpy() @ self.Wt)))

    def embed_video(self, f):
        return torch.from_numpy(O.l2norm(np.tanh(f['x'].numpy() @ self.Wv)))

    def pack(self, E, layer=None):
        E = E.reshape(E.shape[0], -1)
        h = E.to(torch.float16).contiguous()
        return Packed(h.view(torch.uint8).reshape(-1), E.shape[0], E.shape[1])

    def pack_gathered(self, E):
        return self.pack(E)

    @staticmethod
    def _mat(p):
        return p.buf[:p.N * p.K * 2].view(torch.float16).reshape(p.N, p.K).float()

    def sim(self, T, V, heads):
        return self._mat(T) @ self._mat(V).T

    class State:
        pass

    def prepare(self, Et, Ev, T, V, gt, col0):
        """exact ground-truth scores (float64) of the texts whose video is in this shard, -inf elsewhere"""
        st = self.State()
        st.Et, st.Ev, st.T, st.V, st.gt, st.col0 = Et, Ev, T, V, gt, col0
        st.S64 = torch.from_numpy(O.txt2vis_matrix_f64(Et.reshape(Et.shape[0], -1).numpy(), Ev.reshape(Ev.shape[0], -1).numpy()))
        c = gt.long() - col0
        ok = (c >= 0) & (c < st.S64.shape[1])
        st.s_gt64 = torch.full((st.S64.shape[0],), float('-inf'), dtype=torch.float64)
        st.s_gt64[ok] = st.S64[torch.arange(st.S64.shape[0])[ok], c[ok]]
        return st

    def s_gt_of(self, st):
        return st.s_gt64

    def sim_ranked(self, st, want_scores=True):
        """fp16-operand score block + counts of the EXACT scores above the exact ground-truth score"""
        S = self.sim(st.T, st.V, 1)
        cols = torch.arange(S.shape[1])[None, :] + st.col0
        count = ((st.S64 > st.s_gt64[:, None]) & (cols != st.gt.long()[:, None])).sum(dim=1).to(torch.int32)
        return S, count

    def metrics(self, ranks):
        r = ranks.numpy().astype(np.float64)
        return O.eval_from_positions([[x] for x in r])


def _problem(Nt=61, Nv=23, D=32, seed=5):
    g = np.random.default_rng(seed)
    zv = g.normal(0, 1, (Nv, 8)).astype(np.float32)
    gt = (np.arange(Nt) % Nv).astype(np.int32)
    xv = (zv @ g.normal(0, 1, (8, 16)) + 0.3 * g.normal(0, 1, (Nv, 16))).astype(np.float32)
    xt = (zv[gt] @ g.normal(0, 1, (8, 16)) + 0.3 * g.normal(0, 1, (Nt, 16))).astype(np.float32)
    Wt = g.normal(0, 0.3, (16, D)).astype(np.float32)
    Wv = g.normal(0, 0.3, (16, D)).astype(np.float32)
    return xt, xv, gt, Wt, Wv


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
