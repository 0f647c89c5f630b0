"""Fusion blocks of the reference's model/Attention.py that lie on the hot path, same class names, constructor
arguments, parameter names (state_dict keys) and output shapes; forward() runs the `laff_fuse` HIP kernel.

  Attention_1                    /root/reference/model/Attention.py:40-105
  Multi_head_MyApply_Attention   /root/reference/model/Attention.py:473-552
  JustAverage                    /root/reference/model/Attention.py:26-37

The other 13 variants of that file are ablation baselines never selected by the shipped scripts
 and are not provided.
"""
import torch
import torch.nn as nn

from .. import ops


def _planes_of(local_embs):
    """(N, L, D) stacked tensor -> list of per-feature strided row views (no copy)."""
    if local_embs.dim() != 3:
        raise ValueError('local_embs must be (batch, L, embed_dim)')
    if local_embs.stride(2) != 1:
        local_embs = local_embs.contiguous()
    return [(local_embs[:, l, :], False, None, None) for l in range(local_embs.shape[1])]


def _full_width(plane, heads, D):
    """a tiled plane (src (N, D / heads), tile, scale, shift) written out as (N, D): repeat over heads + the folded affine"""
    out = ops.fuse([plane], heads, D // heads, None, None, None, ops.attention_flags(just_average=True))
    return out.view(out.shape[0], D)


class JustAverage(nn.Module):
    def forward(self, local_embs, raw_global_emb=None):
        return self.fuse_planes(_planes_of(local_embs))

    def fuse_planes(self, planes, heads=1, l2norm_planes=False):
        H = heads if any(p[1] for p in planes) else 1
        D = planes[0][0].shape[1] * (heads if planes[0][1] else 1)
        out = ops.fuse(planes, H, D // H, None, None, None, ops.attention_flags(just_average=True), l2norm_planes=l2norm_planes)
        return out.view(out.shape[0], D)


class Attention_1(nn.Module):
    """softmax_L(Linear(d,1)(c)) weighted sum over the L fused features (+ gw * mean), then l2norm(eps=0)."""

    def __init__(self, embed_dim, with_ave=True, mul=False):
        super().__init__()
        self.with_ave = with_ave
        self.mul = mul
        self.embed_dim = embed_dim
        self.embedding_common = nn.Sequential(nn.Linear(embed_dim, 1))
        self.weights = 0
        self.global_emb_weight_net = nn.Linear(1, 1, False)
        self.change_raw_global_emb_weight(1)

    def get_raw_global_emb_weight(self):
        return self.global_emb_weight_net.weight.item()

    def change_raw_global_emb_weight(self, new_value):
        self.global_emb_weight_net.weight.data.fill_(new_value)

    def get_attention_weight(self):
        return torch.as_tensor(self.weights).clone().detach().cpu()

    def _params(self):
        lin = self.embedding_common[0]
        return (lin.weight.detach().reshape(1, -1).contiguous(), lin.bias.detach().reshape(1).contiguous(),
                self.global_emb_weight_net.weight.detach().reshape(1).contiguous())

    def fuse_planes(self, planes, heads=1, l2norm_planes=False, record_weights=None):
        """record_weights: also store the softmax weights (the reference's `self.weights` side output, Attention.py:90,97).  The
        tower path leaves it off unless `self.record_weights` is set (get_attention_weight does): it is an extra N x L store per
        launch that only that consumer reads."""
        if self.training:
            raise NotImplementedError('laff_amd implements the inference path only; call .eval()')
        w, b, gw = self._params()
        flags = ops.attention_flags(self.with_ave, self.mul)
        if heads > 1 and any(p[1] for p in planes):
            # a no-transform feature repeated over `heads` (model/model.py:1822-1823) in front of a single-head attention: this block
            # sees all heads * d columns as one vector, so the tiled plane is materialised to full width first
            planes = [(_full_width(p, heads, self.embed_dim), False, None, None) if p[1] else p for p in planes]
        packed = getattr(self, 'emit_packed', None)       # 'fp16' | 'bf16': also emit the GEMM operand (last_packed)
        rec = getattr(self, 'record_weights', False) if record_weights is None else record_weights
        # rank_side (ops.FusedPrepare.video / .text, set for one pass by its owner): laff_rank_prepare's work for these rows rides along
        side = getattr(self, 'rank_side', None) if packed else None
        res = ops.fuse(planes, 1, self.embed_dim, w, b, gw, flags, return_weights=rec, packed_precision=packed,
                       l2norm_planes=l2norm_planes, rank_side=side)
        res = res if isinstance(res, tuple) else (res,)
        E = res[0]
        self.last_packed = res[-1] if packed else None
        if rec:
            aw = res[1][:, 0, :]
            if self.with_ave:   # what the reference stashes in that case (Attention.py:97)
                aw = aw + gw / aw.shape[1]
            self.weights = aw
        return E.view(E.shape[0], self.embed_dim)

    def forward(self, local_embs, raw_global_emb=None):
        if raw_global_emb is not None:
            raise NotImplementedError('raw_global_emb is never passed on the retrieval path '
                                      '(and is undefined in the reference when mul=False)')
        return self.fuse_planes(_planes_of(local_embs), record_weights=True)


class Multi_head_MyApply_Attention(nn.Module):
    """H independent Attention_1 blocks over head slices of the common space -> (batch, H, dim_per_head)."""

    def __init__(self, embed_dim, multi_heads=None, dim_per_head=None, with_ave=True, mul=True, split_head=True,
                 l2norm_each_head=False):
        super().__init__()
        if embed_dim is None:
            return
        self.dim_per_head = dim_per_head
        self.multi_heads = multi_heads
        self.split_head = split_head
        if self.split_head:
            assert dim_per_head == embed_dim // multi_heads
        else:
            dim_per_head = embed_dim
        self.head_dim = dim_per_head
        self.with_ave, self.mul = with_ave, mul
        self.attention_layer = nn.Sequential()
        for i in range(multi_heads):
            self.attention_layer.add_module(str(i), Attention_1(dim_per_head, with_ave=with_ave, mul=mul))
        self.layer_norm = nn.LayerNorm(dim_per_head)   # declared and unused, as in the reference (:504)
        self.l2norm_each_head = l2norm_each_head
        self._packed = None

    def _params(self):
        ps = [p for h in range(self.multi_heads) for p in self.attention_layer[h].parameters()]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or self._packed[0] != key:
            w = torch.stack([self.attention_layer[h].embedding_common[0].weight.detach().reshape(-1)
                             for h in range(self.multi_heads)]).contiguous()
            b = torch.cat([self.attention_layer[h].embedding_common[0].bias.detach().reshape(1)
                           for h in range(self.multi_heads)]).contiguous()
            gw = torch.cat([self.attention_layer[h].global_emb_weight_net.weight.detach().reshape(1)
                            for h in range(self.multi_heads)]).contiguous()
            self._packed = (key, w, b, gw)
        return self._packed[1:]

    def fuse_planes(self, planes, heads=None, l2norm_planes=False, record_weights=None):
        if self.training:
            raise NotImplementedError('laff_amd implements the inference path only; call .eval()')
        w, b, gw = self._params()
        flags = ops.attention_flags(self.with_ave, self.mul, self.l2norm_each_head, self.split_head)
        packed = getattr(self, 'emit_packed', None)       # 'fp16' | 'bf16': also emit the GEMM operand (last_packed)
        rec = getattr(self, 'record_weights', False) if record_weights is None else record_weights
        side = getattr(self, 'rank_side', None) if packed else None
        res = ops.fuse(planes, self.multi_heads, self.head_dim, w, b, gw, flags, return_weights=rec, packed_precision=packed,
                       l2norm_planes=l2norm_planes, rank_side=side)
        res = res if isinstance(res, tuple) else (res,)
        E = res[0]
        self.last_packed = res[-1] if packed else None
        if rec:
            aw = res[1]
            for h in range(self.multi_heads):
                a = aw[:, h, :]
                self.attention_layer[h].weights = a + gw[h] / a.shape[1] if self.with_ave else a
        return E

    def forward(self, local_embs, raw_global_emb=None, attn_mask=None):
        return self.fuse_planes(_planes_of(local_embs), record_weights=True)

    def get_raw_global_emb_weight(self):
        return self.attention_layer[0].global_emb_weight_net.weight.item()

    def change_raw_global_emb_weight(self, new_value):
        for i in range(self.multi_heads):
            self.attention_layer[i].global_emb_weight_net.weight.data.fill_(new_value)

    def get_attention_weight(self, head=0):
        return self.attention_layer[head].get_attention_weight().detach()
