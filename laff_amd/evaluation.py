"""evaluation.py of the reference (/root/reference/evaluation.py) for the hot path.

`eval`, `eval_qry2retro`, `l2norm`, `cosine_sim` keep the reference signatures (host numpy in, host numpy / tuple
out).  The retrieval pipeline itself never builds a label matrix: it feeds integer ranks produced on the GPU
(laff_rank_count / laff_v2t_count) to `eval_from_positions`, which computes the same seven numbers.
"""
import numpy as np
import torch

from . import ops


def l2norm(X):
    """evaluation.py:11-16 (host numpy)."""
    norm = np.linalg.norm(X, axis=1, keepdims=True)
    return 1.0 * X / (norm + 1e-10)


def cosine_sim(query_embs, retro_embs, precision='fp16x3', device='cuda'):
    """evaluation.py:44-49: l2norm (eps 1e-10) both sides then dot; computed on the GPU, returned as numpy."""
    q = torch.as_tensor(np.ascontiguousarray(query_embs, dtype=np.float32), device=device)
    r = torch.as_tensor(np.ascontiguousarray(retro_embs, dtype=np.float32), device=device)
    # loss-style l2norm adds eps + 1e-14; eps = 1e-10 - 1e-14 reproduces the numpy denominator
    Q = ops.pack_rows(q, True, 1e-10 - 1e-14, precision)
    R = ops.pack_rows(r, True, 1e-10 - 1e-14, precision)
    return ops.sim_gemm(Q, R).cpu().numpy()


def compute_sim(query_embs, retro_embs, measure='cosine', device='cuda'):
    if measure != 'cosine':
        raise NotImplementedError("measure '%s' is never configured on the path" % measure)
    return cosine_sim(query_embs, retro_embs, device=device)


def eval_from_positions(first_rank, ap):
    """The arithmetic of evaluation.eval (:103-109) given per-row first GT rank (1-based) and AP."""
    ranks = np.asarray(first_rank, dtype=np.float64)
    r1, r5, r10 = [100.0 * np.mean(ranks <= k) for k in (1, 5, 10)]
    medr = np.floor(np.median(ranks))
    meanr = ranks.mean()
    mir = (1.0 / ranks).mean()
    mAP = np.asarray(ap, dtype=np.float64).mean()
    return (r1, r5, r10, medr, meanr, mir, mAP)


def eval(label_matrix):
    """evaluation.py:92-109 on a 0/1 label matrix whose columns are in ranked order."""
    lab = np.asarray(label_matrix).astype(int) == 1
    if not lab.any(axis=1).all():
        raise IndexError('a row of label_matrix has no ground truth')   # the reference fails on rank[0]
    first = lab.argmax(axis=1) + 1
    pos = np.arange(1, lab.shape[1] + 1, dtype=np.float64)
    hits = np.cumsum(lab, axis=1)
    ap = (np.where(lab, hits / pos, 0.0).sum(axis=1)) / lab.sum(axis=1)
    return eval_from_positions(first, ap)


def eval_qry2retro(qry2retro_sim, n_qry=1):
    """evaluation.py:64-89 (0-based ranks; GT of query i is column i/n_qry, meaningful for n_qry == 1)."""
    sim = np.asarray(qry2retro_sim)
    assert sim.shape[0] / sim.shape[1] == n_qry, sim.shape
    if n_qry != 1:
        raise NotImplementedError('the reference compares against index/n_qry with true division: only n_qry=1 works')
    diag = sim[np.arange(sim.shape[0]), np.arange(sim.shape[0])]
    ranks = (sim > diag[:, None]).sum(axis=1).astype(np.float64)
    r1 = 100.0 * np.sum(ranks < 1) / len(ranks)
    r5 = 100.0 * np.sum(ranks < 5) / len(ranks)
    r10 = 100.0 * np.sum(ranks < 10) / len(ranks)
    return (r1, r5, r10, np.floor(np.median(ranks)) + 1, ranks.mean() + 1, (1.0 / (ranks + 1)).mean())
