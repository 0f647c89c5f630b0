"""Device-resident evaluation pipeline: features in HBM -> embeddings -> score matrix -> ranks -> 7 metrics.

This is do_predictor's hot path (model.predict + the rank/label loop + evaluation.eval, /root/reference/predictor.py:224-246)
with every intermediate kept in HBM; it is what bench.py times.
"""
import torch

from . import ops


class RetrievalResult:
    def __init__(self, S, ranks, metrics, vis_emb, txt_emb):
        self.S, self.ranks, self.metrics, self.vis_emb, self.txt_emb = S, ranks, metrics, vis_emb, txt_emb


def embed(model, vis_feats, txt_feats):
    """Both towers on whole feature matrices (one FC GEMM per feature, one fuse launch per side)."""
    vis = dict(vis_feats)
    frame_dict = {}
    if 'mask_tensor' in vis:                      # FrameLAFF workload: every video feature is a frame tensor
        frame_dict, vis = vis, {}
    vis_emb = model.vis_net(vis, vis_frame_feat_dict_input=frame_dict)
    cap = dict(txt_feats)
    cap.setdefault('caption', None)
    txt_emb = model.txt_net(cap)
    return vis_emb, txt_emb


def evaluate(model, vis_feats, txt_feats, gt, precision='fp16', write_scores=True, fused_rank=False, want_metrics=True):
    """One pass of the hot path.  gt: int32 (Nt,) device tensor of ground-truth video columns."""
    with torch.no_grad():
        vis_emb, txt_emb = embed(model, vis_feats, txt_feats)
        heads = vis_emb.shape[1] if vis_emb.dim() == 3 else 1
        T = ops.pack_rows(txt_emb, True, 1e-13, precision)
        V = ops.pack_rows(vis_emb, True, 1e-13, precision)
        if fused_rank and not write_scores:
            raise NotImplementedError('rank-only mode needs the ground-truth pre-pass (round 2)')
        S = ops.sim_gemm(T, V, heads=heads)
        s_gt = ops.gather_gt(S, gt)
        ranks = ops.rank_count(S, gt, s_gt) + 1
        metrics = ops.rank_metrics(ranks) if want_metrics else None
    return RetrievalResult(S, ranks, metrics, vis_emb, txt_emb)
