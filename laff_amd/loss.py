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
