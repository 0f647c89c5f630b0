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
    cap = dict(txt_feats)
    cap.setdefault('caption', None)
    pending = []                                  # all FC projections of both towers -> ONE grouped launch
    fin_v = model.vis_net.prepare(vis, frame_dict, pending)
    fin_t = model.txt_net.prepare(cap, pending)
    from .model.model import run_fc
    run_fc(pending)
    return fin_v(), fin_t()


def embed_split(model, vis_feats, txt_feats):
    """embed() in two steps: every FC projection of both towers (ONE grouped launch) and the VIDEO fusion now; returns (vis_emb, finish_text)
    -- finish_text() launches the text fusion.  For callers that have something to start between the two (the sharded pass sends the
    video rows to the other ranks while the text side is fused)."""
    vis = dict(vis_feats)
    frame_dict = {}
    if 'mask_tensor' in vis:
        frame_dict, vis = vis, {}
    cap = dict(txt_feats)
    cap.setdefault('caption', None)
    pending = []
    fin_v = model.vis_net.prepare(vis, frame_dict, pending)
    fin_t = model.txt_net.prepare(cap, pending)
    from .model.model import run_fc
    run_fc(pending)
    return fin_v(), fin_t


def evaluate(model, vis_feats, txt_feats, gt, precision='fp16', write_scores=True, want_metrics=True, exact=True):
    """One pass of the hot path on one GPU.  gt: int32 (Nt,) device tensor of ground-truth video columns.
    write_scores=False skips materialising S (ranks only).

    exact=True (default): ranks are those of the exact cosine scores of the fp32 embeddings whatever the GEMM operand
    precision (ops.exact_ranks: error-band count in the GEMM epilogue + exact re-score of the pairs inside the band) -- the
    reference ranks on fp32 scores (predictor.py:232-244).  exact=False: ranks of the reduced-precision scores themselves."""
    with torch.no_grad():
        vis_emb, txt_emb = embed(model, vis_feats, txt_feats)
        heads = vis_emb.shape[1] if vis_emb.dim() == 3 else 1
        T = ops.pack_rows(txt_emb, True, 1e-13, precision)
        V = ops.pack_rows(vis_emb, True, 1e-13, precision)
        state = None
        if exact:
            S, count, state = ops.exact_ranks(txt_emb, vis_emb, T, V, gt, write_scores)
        elif precision == 'fp32':
            S = ops.sim_gemm(T, V, heads=heads)
            s_gt = ops.gather_gt(S, gt)
            count = ops.rank_count(S, gt, s_gt)
        else:
            count = torch.empty((T.N,), dtype=torch.int32, device=gt.device)
            s_gt = ops.row_dot_gt(T, V, gt, heads, zero_count=count)
            S = ops.sim_gemm(T, V, heads=heads, want_scores=write_scores, gt_col=gt, s_gt=s_gt, count=count)
        if want_metrics:
            ranks = torch.empty_like(count)
            metrics = ops.rank_metrics(count, base=1, ranks_out=ranks)
        else:
            ranks, metrics = count + 1, None
    res = RetrievalResult(S, ranks, metrics, vis_emb, txt_emb)
    res.rank_state = state
    return res
