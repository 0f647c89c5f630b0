"""The ranking half of the reference's predictor.get_predict_file (/root/reference/predictor.py:232-276) on the GPU.

The reference argsorts the (Nt, Nv) score matrix on the host, builds an (Nt, Nv) float64 label matrix with a
per-query Python loop of numpy string comparisons, and hands it to evaluation.eval -- twice (T2V and V2T).
Here ids are matched once through a dict, ranks are counted on the device straight from the score matrix
(position = 1 + #{strictly greater}), and only Nt integers come back to the host.
"""
import numpy as np
import torch

from . import evaluation, ops


def gt_columns(txt_ids, vis_ids):
    """owner[t] = column of the video named by `txt_id.split('#')[0]` (predictor.py:241).  Large id lists are matched by the library's
    host helper (laff_match_ids: two joined blobs, one hash table -- 40,000 captions against 10,000 videos in ~1.5 ms where the
    per-string Python loop below takes ~5); same results, same exceptions."""
    if len(txt_ids) >= 2048:
        try:
            tb, vb = '\n'.join(txt_ids).encode(), '\n'.join(vis_ids).encode()
        except (TypeError, UnicodeError):
            tb = None
        if tb is not None:
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            out = np.empty(len(txt_ids), dtype=np.int32)
            rc = lib.laff_match_ids(tb, len(tb), len(txt_ids), vb, len(vb), len(vis_ids), out.ctypes.data_as(C.c_void_p))
            if rc == 0:
                return out
            msg = lib.laff_last_error().decode()
            if 'appears twice' in msg:
                raise ValueError(msg)
            if 'refers to a video' in msg:
                raise IndexError(msg)
            # (an id with a line break in it: the blobs do not split into the ids -- the per-string loop below handles them)
    index = {}
    for i, v in enumerate(vis_ids):
        if v in index:
            raise ValueError("video id '%s' appears twice in vis_ids" % v)
        index[v] = i
    try:
        return np.fromiter((index[t.partition('#')[0]] for t in txt_ids), dtype=np.int32, count=len(txt_ids))      # (= t.split('#')[0], half the time)
    except KeyError as e:
        raise IndexError('caption refers to a video that is not in vis_ids: %s' % e)   # reference: gt_index[0] fails


def t2v_ranks(S, owner):
    """1-based rank of the ground-truth video of every text row, counted on the device."""
    gt = torch.as_tensor(owner, dtype=torch.int32, device=S.device)
    s_gt = ops.gather_gt(S, gt)
    return ops.rank_count(S, gt, s_gt) + 1


def t2v_metrics(S, owner):
    """(r1, r5, r10, medr, meanr, mir, mAP) of predictor.py:232-246 (one GT per text => AP = 1/rank)."""
    return ops.rank_metrics(t2v_ranks(S, owner).to(torch.int32))


def v2t_positions(S, owner, state=None):
    """For every text t: 1-based position of t in the descending column of its owner video.  With the RankState of the exact-rank
    pipeline that produced S (ops.exact_ranks) the positions are those of the fp64 scores (laff_v2t_count_exact); without it they
    are counted on S as it is."""
    owner = np.asarray(owner, dtype=np.int64)
    Nv = S.shape[1]
    order = np.argsort(owner, kind='stable').astype(np.int32)
    counts = np.bincount(owner, minlength=Nv)
    off = np.zeros(Nv + 1, dtype=np.int32)
    np.cumsum(counts, out=off[1:])
    d_off, d_order = torch.as_tensor(off, device=S.device), torch.as_tensor(order, device=S.device)
    if state is not None:
        cnt = ops.v2t_count_exact(S, state, d_off, d_order, int(counts.max()))
    else:
        cnt = ops.v2t_count(S, d_off, d_order, int(counts.max()))
    return cnt.cpu().numpy().astype(np.int64) + 1, order, off


def v2t_metrics(S, owner, state=None):
    """predictor.py:262-276: per video, positions of all its captions -> evaluation.eval arithmetic."""
    pos, order, off = v2t_positions(S, owner, state)
    Nv = S.shape[1]
    if (np.diff(off) == 0).any():
        raise IndexError('a video has no caption: the reference fails on rank[0] (evaluation.py:99)')
    p = pos[order]                                   # grouped by video
    grp = np.repeat(np.arange(Nv), np.diff(off))
    # sort positions inside each group; equal scores among a video's own captions get consecutive places
    key = np.lexsort((p, grp))
    p, grp = p[key], grp[key]
    start = off[:-1][grp]
    within = np.arange(len(p)) - start               # 0-based index i of the GT inside its group
    for _ in range(int(np.diff(off).max())):         # make ties strictly increasing
        bump = (within > 0) & (p <= np.roll(p, 1))
        if not bump.any():
            break
        p = np.where(bump, np.roll(p, 1) + 1, p)
    first = p[off[:-1]]
    ap = np.add.reduceat((within + 1.0) / p, off[:-1]) / np.diff(off)
    return evaluation.eval_from_positions(first, ap)


def retrieval_metrics(S, txt_ids, vis_ids, state=None):
    """Both directions from a device score matrix, as get_predict_file reports them.  state: the RankState of the exact-rank pipeline
    that produced S (ops.exact_ranks / model.retrieve); with it both directions are ranked on the fp64 scores."""
    if not isinstance(S, torch.Tensor):
        S = torch.as_tensor(np.ascontiguousarray(S, dtype=np.float32), device='cuda')
    owner = gt_columns(txt_ids, vis_ids)
    if state is not None:
        if not np.array_equal(state.gt_col.cpu().numpy(), owner):
            raise ValueError('state was prepared for other ground-truth columns than txt_ids / vis_ids give')
        t2v = ops.rank_metrics(state.count, base=1)
        return t2v, v2t_metrics(S, owner, state)
    return t2v_metrics(S, owner), v2t_metrics(S, owner)


def topk_lists(S, vis_ids, Threshold=2000, block_rows=None):
    """(idx (Nt,K) int32, val (Nt,K) float32) numpy: the ranked lists the reference's writers keep per query
    (predictor.py:55-65): the best `Threshold` videos when the collection has at least that many, else -- faithfully to
    the reference's `inds[index][::-1][0:-1]` -- all but the last one.  Selected and sorted on the device.
    S: the (Nt, Nv) score matrix, or `(T, V, heads)` -- the packed GEMM operands (ops.pack_rows) -- in which case the matrix is
    never materialised (ops.topk_from_operands: blocks of texts scored and reduced one after the other)."""
    from_ops = isinstance(S, tuple) and len(S) == 3 and isinstance(S[0], ops.Packed)
    if not from_ops and not isinstance(S, torch.Tensor):
        S = torch.as_tensor(np.ascontiguousarray(S, dtype=np.float32), device='cuda')
    Nv = len(vis_ids)
    Nt = S[0].N if from_ops else S.shape[0]
    K = Threshold if Nv >= Threshold else Nv - 1
    if K < 1:
        return np.zeros((Nt, 0), np.int32), np.zeros((Nt, 0), np.float32)
    if from_ops:
        idx, val = ops.topk_from_operands(S[0], S[1], K, heads=S[2], block_rows=block_rows)
    else:
        idx, val = ops.topk_rows(S, K)
    return idx.cpu().numpy(), val.cpu().numpy()


def txt2video_write_to_file(pred_result_file, S, vis_ids, txt_ids, pkl_saved_file=None, txt_loader=None, Threshold=2000, block_rows=None):
    """predictor.txt2video_write_to_file (predictor.py:53-88) fed by the device top-K instead of a full-matrix argsort:
    `id.sent.score.txt` lines `txt_id vis_id score vis_id score ...` (scores printed as numpy float32, like the
    reference) and the `t2v.pkl` dict {txt_id: {query, rank_list, sim_value}}."""
    import pickle
    idx, val = topk_lists(S, vis_ids, Threshold, block_rows)       # (S may be the packed operands (T, V, heads): no score matrix)
    vis = np.asarray(vis_ids, dtype=object)
    shot_dict = {}
    fout = open(pred_result_file, 'w') if pred_result_file is not None else None
    try:
        for r in range(idx.shape[0]):
            names = vis[idx[r]]
            if fout is not None:
                fout.write(txt_ids[r] + ' ' + ' '.join([n + ' %s' % v for n, v in zip(names, val[r])]) + '\n')
            if pkl_saved_file is not None:
                shot_dict[txt_ids[r]] = {
                    'query': txt_loader.dataset.get_caption_dict_by_id(txt_ids[r])['caption'] if hasattr(txt_loader.dataset, 'get_caption_dict_by_id')
                    else txt_loader.dataset.captions[txt_ids[r]],
                    'rank_list': list(names), 'sim_value': list(val[r])}
    finally:
        if fout is not None:
            fout.close()
    if pkl_saved_file is not None:
        with open(pkl_saved_file, 'wb') as f:
            pickle.dump(shot_dict, f)
    return idx, val
