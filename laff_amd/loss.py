"""loss.l2norm / loss.cosine_sim of the reference (/root/reference/loss.py:8-34) on the HIP kernels."""
import torch

from . import ops

#: operand precision of the similarity GEMM; 'fp16' meets the 1e-4 cosine contract, 'fp16x3' is ~1e-7.
DEFAULT_PRECISION = 'fp16'


def l2norm(X, eps=1e-13, dim=1):
    """X / (sqrt(sum X^2 along dim) + eps + 1e-14); dim must be the last axis of a 2-D/3-D tensor."""
    if X.dim() == 2 and dim in (1, -1):
        return ops.pack_rows(X.contiguous(), True, eps, 'fp32', 1.0).buf.view(torch.float32).view(X.shape)
    if X.dim() == 3 and dim in (2, -1):
        return ops.pack_rows(X.contiguous(), True, eps, 'fp32', 1.0).buf.view(torch.float32).view(X.shape)
    raise NotImplementedError('l2norm along dim=%d of a %d-D tensor is outside the hot path' % (dim, X.dim()))


def cosine_sim(query, retrio, precision=None):
    """l2norm(query) @ l2norm(retrio).T  (loss.py:30-34): re-normalises both operands like the reference."""
    precision = precision or DEFAULT_PRECISION
    q = ops.pack_rows(query.contiguous(), True, 1e-13, precision)
    r = ops.pack_rows(retrio.contiguous(), True, 1e-13, precision)
    return ops.sim_gemm(q, r, heads=1)


# ---- training loss ---------------------------------------------------------------------------
class _MarginRankingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, im, margin, max_violation, cost_style, direction):
        need = s.requires_grad or im.requires_grad
        loss, d_s, d_im = ops.margin_loss(s.detach(), im.detach(), margin, max_violation, cost_style, direction, want_grad=need)
        ctx.save_for_backward(d_s, d_im) if need else None
        ctx.has_grad = need
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.has_grad:
            return (None,) * 6
        d_s, d_im = ctx.saved_tensors
        return grad_out * d_s, grad_out * d_im, None, None, None, None


class MarginRankingLoss(torch.nn.Module):
    """loss.MarginRankingLoss (/root/reference/loss.py:68-135) with measure='cosine': forward(s, im) on (B, d) inputs as the
    reference, or on (B, H, d) for the per-head sum of model/model.py:2032-2048 in one call.  Forward and backward run in
    liblaff_hip.so (laff_margin_loss); the module plugs into autograd."""

    def __init__(self, margin=0, measure='cosine', max_violation=False, cost_style='sum', direction='bidir', device=None):
        super().__init__()
        if measure != 'cosine':
            raise NotImplementedError("only measure='cosine' is on the path ('hist' is the Jaccard variant of task 2)")
        self.margin, self.max_violation, self.cost_style, self.direction = margin, max_violation, cost_style, direction

    def forward(self, s, im):
        return _MarginRankingFn.apply(s, im, self.margin, self.max_violation, self.cost_style, self.direction)


def compute_loss(criterion, vis_embs, txt_embs):
    """W2VVPP_MultiHeadAttention.compute_loss with multi_space=True (model/model.py:2032-2048): one criterion per head, summed.
    Returns (loss, {'triplet_loss': loss})."""
    if vis_embs.dim() != txt_embs.dim() or vis_embs.dim() not in (2, 3):
        raise Exception('vis_embs dims are not equal to txt_embs dims')
    loss = criterion(txt_embs, vis_embs)
    return loss, {'triplet_loss': loss}
