"""Retrieval towers and predict() of the reference's model/model.py on the MI355X kernels.

Same registry keys, class names, constructor/forward signatures, config attribute names and state_dict keys
as /root/reference/model/model.py for the symbols on the hot path:

  TransformNet                         :211-276     -> laff_fc_act_bn (fp32 MFMA GEMM + fused epilogue)
  VisMutiTransformNet                  :1787-1827
  VisMutiTransformNetAddAttnetion      :1830-1881   -> planes + one laff_fuse launch (no stack / repeat copies)
  MultiScaleTxtEncoderAttention        :1641-1709
  VisMutiTransformNetPlusFrameFeat     :2101-2194   -> laff_frame_fuse instead of the per-sample Python loop
  W2VVPP.get_txt2vis_matrix / predict  :1003-1128   -> one laff_sim_gemm over all pairs instead of the
                                                       (Nt/bs)x(Nv/bs) block loop; embeddings stay in HBM
  get_model                            :2501-2519

Inference only: training (forward(), losses, optimisers) is out of scope and raises.
Text encoders (GRU / BoW / W2V / CLIP, :311-549) are upstream of the path: features arrive pre-extracted in
`caption_feat_dict` (the reference already does this for frozen CLIP via 'CLIP_encoding', :497-498); any module
returning {'text_features': tensor} can be plugged into `txt_net.encoder.<name>`.
"""
import numpy as np
import os

import torch
import torch.nn as nn

from .. import loss as _loss
from .. import ops
from .Attention import Attention_1, JustAverage, Multi_head_MyApply_Attention

device = torch.device('cuda')
float16 = False
#: arithmetic of the FC projections: 'fp32' (v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 FMA chain) or 'fp16x3'
#: (exact fp16 hi/lo operand split, 3 MFMA passes at the fp16 rate, ~2^-22 relative per product)
FC_PRECISION = 'fp32'
#: tower path: sparse (bag-of-words) features are gathered inside the fuse launch (LAFF_GATHER_IN_FUSE=0: separate FC launch)
GATHER_IN_FUSE = os.environ.get('LAFF_GATHER_IN_FUSE', '1') != '0'
#: fp16x3 only: split the inputs inside the GEMM instead of materialising their hi/lo planes (LAFF_FUSED_SPLIT=0 disables)
FUSED_SPLIT = os.environ.get('LAFF_FUSED_SPLIT', '1') != '0'
#: fp16x3 only: projections of 512-d inputs take the strip form (X stationary in registers: no row-scale pass, no split planes;
#: laff_fc_act_bn_strip_grouped).  LAFF_FC_STRIP=0 keeps the tiled kernels.
FC_STRIP = os.environ.get('LAFF_FC_STRIP', '1') != '0'


def _row_chunks(pending, limit_bytes=1 << 31):
    """The split-product GEMMs address an operand with 32-bit byte offsets (the C ABI refuses a packed operand of 4 GiB or more:
    300k x 4096 fp32 rows already exceed it).  Row blocks are independent, so an oversized problem is cut into row chunks that share
    its weights and write into views of its output -- same results, a few more tiles in the grouped launch."""
    out = []
    for q in pending:
        x = q['x']
        if not torch.is_tensor(x) or x.layout != torch.strided or x.dim() != 2:
            out.append(q)
            continue
        n, k = x.shape
        row_bytes = 4 * max(x.stride(0) if n > 1 else k, (k + 63) // 64 * 64)
        if n * row_bytes < limit_bytes:
            out.append(q)
            continue
        if q.get('out') is None:
            q['out'] = torch.empty((n, q['weight'].shape[0]), device=x.device, dtype=torch.float32)
        step = max(256, (limit_bytes // row_bytes) // 256 * 256)
        for a in range(0, n, step):
            out.append(dict(q, x=x[a:a + step], out=q['out'][a:a + step]))
    return out


def _loader_len(loader):
    ds = getattr(loader, 'dataset', None)
    try:
        return len(ds) if ds is not None else len(loader)
    except TypeError:
        return 1


def coalesce_batches(batches):
    """Concatenate the batch dicts of a loader into one: tensors / arrays along dim 0, lists and tuples end to end, nested dicts
    key by key, None stays None.  Raises TypeError / ValueError / RuntimeError when the batches do not line up."""
    if not batches:
        return None
    first = batches[0]
    if isinstance(first, dict):
        if any(not isinstance(b, dict) or b.keys() != first.keys() for b in batches):
            raise ValueError('batch dicts with different keys')
        return {k: coalesce_batches([b[k] for b in batches]) for k in first}
    if first is None:
        if any(b is not None for b in batches):
            raise ValueError('None mixed with values')
        return None
    if isinstance(first, torch.Tensor):
        return torch.cat(list(batches), dim=0)
    if isinstance(first, np.ndarray):
        return np.concatenate(list(batches), axis=0)
    if isinstance(first, (list, tuple)):
        return [x for b in batches for x in b]
    raise TypeError('cannot concatenate batch values of type %s' % type(first).__name__)


def run_fc(pending):
    """Launch every queued FC projection as one grouped GEMM."""
    if FC_PRECISION == 'fp16x3':
        whole = pending
        pending = _row_chunks(pending)
        if len(pending) != len(whole):          # chunked: launch the pieces, hand back one output per original problem
            outs = _run_fc_x3(pending)
            by_id = {id(q): o for q, o in zip(pending, outs)}
            return [q['out'] if q.get('out') is not None else by_id[id(q)] for q in whole]
        return _run_fc_x3(pending)
    if FC_PRECISION != 'fp32':
        raise ValueError("FC_PRECISION must be 'fp32' or 'fp16x3'")
    return ops.fc_act_bn_grouped(pending)


def _run_fc_x3(pending):
    if FC_STRIP:
        strip = [q for q in pending if q.get('strip') is not None and ops.fc_strip_eligible(q['x'], q['weight'].shape[0])]
        if strip:
            rest = [q for q in pending if not any(q is s for s in strip)]
            outs = {id(q): o for q, o in zip(strip, ops.fc_act_bn_strip_grouped([dict(q, strip=q['strip']()) for q in strip]))}
            if rest:
                outs.update({id(q): o for q, o in zip(rest, _run_fc_x3_tiled(rest))})
            return [outs[id(q)] for q in pending]
    return _run_fc_x3_tiled(pending)


def _run_fc_x3_tiled(pending):
    # big launches with a narrow output take the fused split (inputs stay fp32 in HBM, split inside the GEMM: every column
    # tile of a row block repeats the conversion, 2x at D = 512 but 16x at D = 4096, where materialising the planes once is
    # cheaper: C5 33.2 ms fused vs 32.9 ms); small launches take the materialised split, whose 128x128 tiles fill the chip
    tiles = sum(((q['x'].shape[0] + 255) // 256) * ((q['weight_split'].N + 255) // 256) for q in pending)
    if (FUSED_SPLIT and tiles >= 512 and all(q['weight_split'].N <= 1024 for q in pending) and
            all(ops.fused_split_eligible(q['x'], q['weight_split']) for q in pending)):
        return ops.fc_act_bn_fused_grouped(pending)
    return ops.fc_act_bn_split_grouped(pending)


#: tower path: leave activation + BatchNorm of the FC projections to the fuse launch (see TransformNet.plane).  Off by default:
#: measured on C4 the GEMM launch gets 0.045 ms shorter and the fuse launches 0.047 ms longer -- the two v_exp/v_rcp per element
#: run at quarter rate wherever they sit (1.372 ms vs 1.362 ms per step over three A/B pairs on one box).
DEFER_ACTIVATION = os.environ.get('LAFF_DEFER_ACT', '0') == '1'


def _initialize_weights(m):
    """Xavier init of Linear layers, identity BatchNorm (model/model.py:51-60)."""
    if type(m) == nn.Linear:
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif type(m) == nn.BatchNorm1d:
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


def to_device_and_float16(x):
    """model/model.py:63-67: despite its name the reference never casts to fp16."""
    return x.to(device=device, dtype=torch.float32)


def _eval_only(module):
    if module.training:
        raise NotImplementedError('laff_amd implements the inference path only; call .eval() '
                                  '(training is out of scope, SURVEY.md section 8f)')


# ------------------------------------------------------------------------------------------------------
# attention registry (model/model.py:70-208) -- only the variants on the path
# ------------------------------------------------------------------------------------------------------
_ATTENTION_1 = {  # name -> (with_ave, mul)   (model/model.py:94-97)
    'attention_noAverageMul_Ave': (True, False),
    'attention_noAveNoAverageMul': (False, False),
    'attention_averageMul': (True, True),
    'average_AverageMul_noAve': (False, True),
}


def get_attention_layer(attention_type, common_space_dim, encoder_num, opt):
    if attention_type in _ATTENTION_1:
        with_ave, mul = _ATTENTION_1[attention_type]
        return Attention_1(common_space_dim, with_ave=with_ave, mul=mul)
    if attention_type == 'just_average':
        return JustAverage()
    if attention_type == 'Multi_head_MyApply_Attention':
        heads = opt.multi_head_attention['heads']
        return Multi_head_MyApply_Attention(
            common_space_dim, heads, common_space_dim // heads,
            with_ave=opt.attention_param_each_head['with_ave'], mul=opt.attention_param_each_head['mul'],
            split_head=opt.attention_param_each_head['split_head'], l2norm_each_head=opt.attention_l2norm)
    raise NotImplementedError("attention type '%s' is an ablation variant outside the LAFF hot path" % attention_type)


# ------------------------------------------------------------------------------------------------------
# a1/a2: TransformNet
# ------------------------------------------------------------------------------------------------------
class TransformNet(nn.Module):
    """fc -> activation -> dropout (identity in eval) -> BatchNorm1d, as one GEMM with a fused epilogue."""

    def __init__(self, fc_layers, opt=None, dropout=None, batch_norm=None, activation=None, fc=True):
        super().__init__()
        if opt is not None:
            if batch_norm is None:
                batch_norm = opt.batch_norm
            if activation is None:
                activation = opt.activation
            if dropout is None:
                dropout = opt.dropout
        self.fc1 = nn.Linear(fc_layers[0], fc_layers[1]) if fc else None
        self.bn1 = nn.BatchNorm1d(fc_layers[1]) if batch_norm else None
        self.activation_name = activation if activation in ('tanh', 'relu', 'sigmoid') else None
        self.dropout_p = dropout if (dropout is not None and dropout > 1e-3) else None
        self.out_features = fc_layers[1]
        self._bn_cache = None
        self.apply(_initialize_weights)

    def bn_affine(self, extra_shift=None):
        """Folded eval-mode BatchNorm1d: y*scale + shift (what aten's inference kernel computes)."""
        if self.bn1 is None:
            if extra_shift is None:
                return None, None
            return torch.ones_like(extra_shift), extra_shift.contiguous()
        bn = self.bn1
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._bn_cache is None or self._bn_cache[0] != key:
            scale = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).contiguous()
            shift = (bn.bias.detach() - bn.running_mean * scale).contiguous()
            self._bn_cache = (key, scale, shift)
        scale, shift = self._bn_cache[1:]
        if extra_shift is not None:
            shift = (shift + extra_shift).contiguous()
        return scale, shift

    def weight_t(self):
        """fc1.weight transposed ([D_k, D]: one vocabulary entry = one contiguous row), cached until the weight changes."""
        w = self.fc1.weight
        key = (w.data_ptr(), w._version)
        if getattr(self, '_w_t', None) is None or self._w_t[0] != key:
            self._w_t = (key, w.detach().t().contiguous())
        return self._w_t[1]

    def weight_split(self):
        """fp16 hi/lo split of fc1.weight for FC_PRECISION == 'fp16x3', cached until the weight changes."""
        if FC_PRECISION != 'fp16x3':
            return None
        w = self.fc1.weight
        key = (w.data_ptr(), w._version)
        if getattr(self, '_w_split', None) is None or self._w_split[0] != key:
            self._w_split = (key, ops.split_rows(w.detach()))
        return self._w_split[1]

    def strip_weights(self):
        """fc1 + folded BatchNorm + activation packed for the strip-form FC (laff_fc_strip_pack), cached until a parameter changes."""
        scale, shift = self.bn_affine(None)
        w, b = self.fc1.weight, self.fc1.bias
        key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version),
               None if self.bn1 is None else self._bn_cache[0], self.activation_name)
        if getattr(self, '_w_strip', None) is None or self._w_strip[0] != key:
            self._w_strip = (key, ops.fc_strip_pack(w.detach(), None if b is None else b.detach(), scale, shift, self.activation_name))
        return self._w_strip[1]

    def plane(self, x, heads=1, extra_shift=None, pending=None, head_dim=None):
        """(src, tile, scale, shift) for laff_fuse.  With an FC the projection either runs now or, when `pending`
        (a list) is given, is appended to it so that the caller launches all features' GEMMs as one grouped kernel."""
        _eval_only(self)
        scale, shift = self.bn_affine(extra_shift)
        if self.fc1 is not None and x.layout == torch.sparse_csr:
            # sparse feature (bag-of-words): gather-sum of columns of W instead of a dense N x |vocab| x D GEMM
            bias = self.fc1.bias.detach() if self.fc1.bias is not None else None
            if pending is not None and GATHER_IN_FUSE and head_dim is not None and head_dim <= 512:
                # tower path: the gather runs INSIDE the fuse launch (the projected plane is never written / re-read)
                return (None, False, scale, shift, self.activation_name, (x.to(device), self.weight_t(), bias))
            y = ops.fc_gather_act_bn(x.to(device), self.weight_t(), bias, scale, shift, self.activation_name)
            return (y, False, None, None)
        x = to_device_and_float16(x)
        if self.fc1 is not None:
            prob = dict(x=x, weight=self.fc1.weight.detach(), weight_split=self.weight_split(),
                        bias=self.fc1.bias.detach() if self.fc1.bias is not None else None,
                        bn_scale=scale, bn_shift=shift, activation=self.activation_name)
            if (FC_PRECISION == 'fp16x3' and FC_STRIP and extra_shift is None and not DEFER_ACTIVATION and
                    self.fc1.in_features == 512 and self.out_features % 32 == 0):
                prob['strip'] = self.strip_weights        # packed lazily: only if the launch takes the strip form
            if pending is None or not DEFER_ACTIVATION:
                if pending is None:
                    return (run_fc([prob])[0], False, None, None)
                y = prob['out'] = torch.empty((x.shape[0], self.out_features), device=x.device, dtype=torch.float32)
                pending.append(prob)
                return (y, False, None, None)
            # tower path: the grouped GEMM writes the pre-activation x W^T + b; activation + BatchNorm ride along in the fuse
            # launch (memory-bound, the transcendentals are free there; in the GEMM epilogue they were 17 % of the launch)
            prob.update(bn_scale=None, bn_shift=None, activation=None)
            y = prob['out'] = torch.empty((x.shape[0], self.out_features), device=x.device, dtype=torch.float32)
            pending.append(prob)
            return (y, False, scale, shift, self.activation_name)
        if self.activation_name is not None:
            raise NotImplementedError('activation without fc is never built by the reference towers')
        tile = heads > 1 and x.shape[1] * heads == self.out_features
        return (x, tile, scale, shift)

    def forward(self, input_x):
        src, tile, scale, shift = self.plane(input_x)[:4]
        if scale is None and not tile:
            return src
        out = ops.fuse([(src, tile, scale, shift)], 1, src.shape[1], None, None, None,
                       ops.attention_flags(just_average=True))
        return out.view(out.shape[0], -1)


# ------------------------------------------------------------------------------------------------------
# a3/a4: video tower of 'LAFF' / 'w2vpp_mutivis_attention'
# ------------------------------------------------------------------------------------------------------
def _expert_rows(embedding, n):
    return None if embedding is None else embedding.weight.detach()[:n]


class _recording:
    """attention layers store their softmax weights (the reference's `self.weights`) only while this is active: on the tower path
    it is an extra store per launch that nobody but get_attention_weight() reads"""

    def __init__(self, *layers):
        self.layers = layers

    def __enter__(self):
        for l in self.layers:
            l.record_weights = True

    def __exit__(self, *a):
        for l in self.layers:
            l.record_weights = False


def _fuse(attention_layer, planes, heads, l2norm_planes=False):
    if hasattr(attention_layer, 'fuse_planes'):
        if l2norm_planes:
            return attention_layer.fuse_planes(planes, heads, l2norm_planes=True)
        return attention_layer.fuse_planes(planes, heads)
    raise NotImplementedError('attention layer %s is outside the LAFF hot path' % type(attention_layer).__name__)


def _materialise(plane, heads, D):
    src, tile, scale, shift = plane[:4]
    act = plane[4] if len(plane) > 4 else None
    if scale is None and not tile and act is None:
        return src
    H = heads if tile else 1
    out = ops.fuse([plane], H, D // H, None, None, None, ops.attention_flags(just_average=True))
    return out.view(out.shape[0], D)


class VisMutiTransformNet(nn.Module):
    def __init__(self, opt, space_dict):
        super().__init__()
        if opt is None:
            return
        self.opt = opt
        self.vis_net_space_dict = space_dict
        self.common_space_dim = opt.vis_fc_layers[1]
        for each in space_dict.keys():
            if each not in opt.vis_no_transform:
                self.add_module(each, TransformNet((space_dict[each], opt.vis_fc_layers[1]), opt))
            else:
                self.add_module(each, TransformNet((space_dict[each], opt.vis_fc_layers[1]), None, dropout=None,
                                                   batch_norm=True, activation=False, fc=False))

    def planes(self, vis_input, expert=None, pending=None):
        if self.opt.vis_feat_add_concat and 'vis_feat_add_concat' not in vis_input:
            vis_input['vis_feat_add_concat'] = torch.cat([to_device_and_float16(v) for v in vis_input.values()], dim=1)
        heads = self.opt.multi_head_attention['heads']
        module_dict = dict(self.named_children())
        out = []
        for i, name in enumerate(self.vis_net_space_dict.keys()):
            vis_input[name] = to_device_and_float16(vis_input[name])     # in-place like the reference (:1817)
            h = heads if name in self.opt.vis_no_transform else 1
            out.append(module_dict[name].plane(vis_input[name], h, None if expert is None else expert[i], pending))
        return out

    def forward(self, vis_input, txt_emb=None, vis_frame_feat_dict_input=None):
        heads = self.opt.multi_head_attention['heads']
        pending = []
        planes = self.planes(vis_input, pending=pending)
        run_fc(pending)
        return {name: _materialise(p, heads, self.common_space_dim)
                for name, p in zip(self.vis_net_space_dict.keys(), planes)}


class VisMutiTransformNetAddAttnetion(nn.Module):
    def __init__(self, opt, space_dict):
        super().__init__()
        if opt is None:
            return
        self.opt = opt
        self.vis_net_space_dict = space_dict
        self.common_space_dim = opt.vis_fc_layers[1]
        self.VisMutiTransformNet = VisMutiTransformNet(opt, space_dict)
        self.attention_layer = get_attention_layer(self.opt.vis_attention, self.common_space_dim, len(space_dict), self.opt)
        self.expert_embedding = None
        self.expert_l2Norm = False
        if opt.vis_expert_embedding['expert']:
            self.expert_embedding = nn.Embedding(len(self.vis_net_space_dict), self.common_space_dim)
        if opt.vis_expert_embedding['l2norm']:
            self.expert_l2Norm = True          # l2norm(local_embs, dim=2) before the attention (model/model.py:1855-1856, 1872-1873)

    def prepare(self, vis_input, vis_frame_feat_dict_input=None, pending=None):
        """Queue this tower's FC projections on `pending`; returns the closure that fuses once they have run."""
        _eval_only(self)
        planes = self.VisMutiTransformNet.planes(vis_input, _expert_rows(self.expert_embedding, len(self.vis_net_space_dict)),
                                                 pending)
        return lambda: _fuse(self.attention_layer, planes, self.opt.multi_head_attention['heads'], self.expert_l2Norm)

    def forward(self, vis_input, txt_emb=None, vis_frame_feat_dict_input=None):
        pending = []
        finish = self.prepare(vis_input, pending=pending)
        run_fc(pending)
        return finish()

    def get_attention_weight(self, vis_input, txt_emb=None):
        with _recording(self.attention_layer):
            self.forward(vis_input, txt_emb)
        return self.attention_layer.get_attention_weight()


# ------------------------------------------------------------------------------------------------------
# text tower
# ------------------------------------------------------------------------------------------------------
class PreExtractedEncoder(nn.Module):
    """Returns a pre-extracted text feature matrix from the caption dict (the reference's own convention for
    frozen CLIP, model/model.py:497-498, extended to every text feature)."""

    def __init__(self, key):
        super().__init__()
        self.key = key

    def forward(self, caption_feat_dict, task3=False):
        if self.key not in caption_feat_dict:
            raise KeyError("caption_feat_dict lacks '%s': text features are pre-extracted on this path; plug an encoder "
                           "module into txt_net.encoder to compute them on the fly" % self.key)
        return {'text_features': caption_feat_dict[self.key]}


class MultiScaleTxtEncoderAttention(nn.Module):
    ENCODER_KEYS = {'rnn_encoder': 'rnn_encoding', 'bert_encoder': 'bert_encoding', 'bow_encoder': 'bow_encoding',
                    'w2v_encoder': 'w2v_encoding', 'CLIP_encoder': 'CLIP_encoding', 'NetVLAD_encoder': 'NetVLAD_encoding'}

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        te = opt.text_encoding
        bow, w2v, rnn, bert, clip_, vlad = (te[k]['name'] for k in ('bow_encoding', 'w2v_encoding', 'rnn_encoding',
                                                                     'bert_encoding', 'CLIP_encoding', 'NetVLAD_encoding'))
        rnn = rnn.split('_', 1)[0]
        # encoder order is fixed by the reference (model/model.py:572-613): rnn, bert, bow, w2v, CLIP, NetVLAD
        self.space_dict = {}
        if rnn == 'gru':
            self.space_dict['rnn_encoder'] = opt.rnn_size
        elif rnn == 'bigru':
            self.space_dict['rnn_encoder'] = opt.rnn_size * 2
        if bert != 'noBert':
            self.space_dict['bert_encoder'] = opt.bert_size
        if 'no' not in bow:
            self.space_dict['bow_encoder'] = opt.t2v_bow.ndims
        if 'no' not in w2v:
            self.space_dict['w2v_encoder'] = opt.t2v_w2v.ndims
        if 'no' not in clip_:
            self.space_dict['CLIP_encoder'] = opt.clip_opt['size']
        if 'no' not in vlad:
            self.space_dict['NetVLAD_encoder'] = opt.t2v_w2v.ndims * opt.NetVLAD_opt['num_clusters']
        self.encoder = nn.Module()
        for name in self.space_dict:
            self.encoder.add_module(name, PreExtractedEncoder(self.ENCODER_KEYS[name]))
        self.encoder_name_list = list(self.space_dict.keys())
        self.txt_encoder_num = len(self.encoder_name_list)

        # transform layers (model/model.py:622-681)
        D = opt.txt_fc_layers[1]
        self.transform_layer = nn.Module()
        for name in ('rnn_encoder', 'bert_encoder', 'w2v_encoder', 'bow_encoder', 'CLIP_encoder', 'NetVLAD_encoder'):
            if name not in self.space_dict:
                continue
            dims = (self.space_dict[name], D)
            if name == 'bert_encoder':
                t = TransformNet(dims, None, opt.bert_transform_dropout, opt.bert_transform_batch_norm,
                                 opt.bert_transform_activation)
            elif name == 'CLIP_encoder':
                co = opt.clip_opt
                if 'CLIP_encoder' in opt.txt_no_transform:
                    t = TransformNet(dims, None, co['transform_dropout'], co['transform_batch_norm'], False, False)
                else:
                    t = TransformNet(dims, None, co['transform_dropout'], co['transform_batch_norm'],
                                     co['transform_activation'])
            else:
                t = TransformNet(dims, None, opt.dropout, opt.batch_norm, opt.activation)
            self.transform_layer.add_module(name + '_transform', t)
        self.attention_layer = get_attention_layer(opt.txt_attention, D, self.txt_encoder_num, opt)

        self.expert_embedding = None
        if opt.txt_expert_embedding['expert']:
            self.expert_embedding = nn.Embedding(len(self.space_dict), D)
        self.txt_expert_l2Norm = bool(opt.txt_expert_embedding['l2norm'])      # model/model.py:1660-1661, 1693-1694

    def prepare(self, caption_feat_dict, pending=None, task3=False):
        _eval_only(self)
        heads = self.opt.multi_head_attention['heads']
        expert = _expert_rows(self.expert_embedding, len(self.encoder_name_list))
        planes = []
        for i, name in enumerate(self.encoder_name_list):
            feats = getattr(self.encoder, name)(caption_feat_dict, task3=task3)['text_features']
            h = heads if name in self.opt.txt_no_transform else 1
            planes.append(getattr(self.transform_layer, name + '_transform').plane(
                feats, h, None if expert is None else expert[i], pending, head_dim=self.opt.txt_fc_layers[1] // heads
                if type(self.attention_layer).__name__ == 'Multi_head_MyApply_Attention' and
                self.attention_layer.split_head and not self.txt_expert_l2Norm else None))     # (a row norm needs the projected plane)
        return lambda: _fuse(self.attention_layer, planes, heads, self.txt_expert_l2Norm)

    def forward(self, caption_feat_dict, visual_emb=None, task3=False):
        pending = []
        finish = self.prepare(caption_feat_dict, pending, task3)
        run_fc(pending)
        return finish()

    def get_attention_weight(self, caption_feat_dict, visual_emb=None):
        with _recording(self.attention_layer):
            self.forward(caption_feat_dict, visual_emb)
        return self.attention_layer.get_attention_weight()


# ------------------------------------------------------------------------------------------------------
# a7: FrameLAFF video tower
# ------------------------------------------------------------------------------------------------------
class VisMutiTransformNetPlusFrameFeat(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        # One TransformNet per input feature under the feature's own name (the state_dict keys of the reference: model/model.py:
        # 2101-2128): video-level features and, with frame_feat_input, the per-frame ones (pooled over frames first, frame_vector).
        # A feature listed in vis_no_transform keeps its width: BatchNorm only, no FC, no activation.
        dims = opt.vis_fc_layers[0]
        names = list(dims.keys())
        if opt.frame_feat_input:
            names += [f for f in opt.vid_frame_feats if f not in dims]
        D_out = opt.vis_fc_layers[1]
        self.vis_net_space_dict = space_dict = {f: opt.vis_fc_layers[0][f] for f in names}
        for f, width in space_dict.items():
            if f in opt.vis_no_transform:
                net = TransformNet((width, D_out), None, dropout=None, batch_norm=True, activation=False, fc=False)
            else:
                net = TransformNet((width, D_out), opt)
            self.add_module(f, net)
        D = opt.vis_fc_layers[1]
        self.vis_attention_layer = get_attention_layer(opt.vis_attention, D, len(space_dict), opt)
        self.frame_attention = nn.ModuleDict()
        for each in opt.vid_frame_feats:
            fd = opt.vis_fc_layers[0][each]
            att = get_attention_layer(opt.vis_frame_attention, fd, 1, opt)
            if not isinstance(att, Attention_1):
                raise NotImplementedError('frame attention must be an Attention_1 variant on this path')
            if opt.vis_frame_addFC:
                self.frame_attention[each] = nn.Sequential(nn.Linear(fd, fd), att)
            else:
                self.frame_attention[each] = nn.Sequential(att)

    def frame_vector(self, feat_name, frames, mask_tensor):
        """(B, Fmax, d) zero-padded frames -> (B, d): Attention_1 over frames for every video in one launch.

        Replaces the per-sample Python loop of model/model.py:2167-2173.  The reference's mask slice is a no-op, so
        padded frames take part in softmax / mean; without a frame FC they are exact zeros and are accounted for
        analytically from `lens`; with vis_frame_addFC the Linear turns them into the bias vector and all Fmax
        frames are processed."""
        seq = self.frame_attention[feat_name]
        att = seq[-1]
        frames = to_device_and_float16(frames).contiguous()
        B, Fmax, d = frames.shape
        lens = None
        if len(seq) == 2:
            fc = seq[0]
            frames = ops.fc_act_bn(frames.view(B * Fmax, d), fc.weight.detach(), fc.bias.detach()).view(B, Fmax, d)
        else:
            lens = mask_tensor.to(device=frames.device).sum(dim=1).to(torch.int32).contiguous()
        w, b, gw = att._params()
        return ops.frame_fuse(frames, lens, w.reshape(-1), b, gw, ops.attention_flags(att.with_ave, att.mul))

    def forward(self, vis_input, vis_frame_feat_dict_input, txt_emb=None):
        pending = []
        finish = self.prepare(vis_input, vis_frame_feat_dict_input, pending)
        run_fc(pending)
        return finish()

    def prepare(self, vis_input, vis_frame_feat_dict_input=None, pending=None):
        _eval_only(self)
        if self.opt.frame_feat_with_video_feat is False:
            vis_input = {}
        names = [k for k in vis_frame_feat_dict_input if k != 'mask_tensor']
        for feat_name in names:
            vis_frame_feat_dict_input[feat_name] = to_device_and_float16(vis_frame_feat_dict_input[feat_name])
        shapes = {tuple(vis_frame_feat_dict_input[k].shape) for k in names}
        if len(names) > 1 and len(shapes) == 1 and not self.opt.vis_frame_addFC:
            # every frame feature of the tower in ONE launch (same shape, same attention type, shared lens)
            frames = [vis_frame_feat_dict_input[k].contiguous() for k in names]
            mask = vis_frame_feat_dict_input['mask_tensor'].to(device=frames[0].device)
            lens = None
            if mask.dtype != torch.float32 or mask.dim() != 2 or mask.stride(1) != 1:
                lens, mask = mask.sum(dim=1).to(torch.int32).contiguous(), None      # (any other mask type: the framework sums it)
            atts = [self.frame_attention[k][-1] for k in names]
            params = []
            for att in atts:
                w, b, gw = att._params()
                params.append((w.reshape(-1), b, gw))
            # (a float mask goes to the launch as it is: the launch takes lens = mask.sum(dim=1) itself)
            vecs = ops.frame_fuse_grouped(frames, lens, params, ops.attention_flags(atts[0].with_ave, atts[0].mul), mask=mask)
            for k, v in zip(names, vecs):
                vis_input[k] = v
        else:
            for feat_name in names:
                vis_input[feat_name] = self.frame_vector(feat_name, vis_frame_feat_dict_input[feat_name],
                                                         vis_frame_feat_dict_input['mask_tensor'])
        heads = self.opt.multi_head_attention['heads']
        module_dict = dict(self.named_children())
        planes = []
        for name in vis_input.keys():
            vis_input[name] = to_device_and_float16(vis_input[name])
            h = heads if name in self.opt.vis_no_transform else 1
            planes.append(module_dict[name].plane(vis_input[name], h, None, pending))
        return lambda: _fuse(self.vis_attention_layer, planes, heads)

    def get_attention_weight(self, vis_input, vis_frame_feat_dict_input):
        self.forward(vis_input, vis_frame_feat_dict_input)
        name = [k for k in vis_frame_feat_dict_input.keys() if k != 'mask_tensor'][0]
        return self.frame_attention[name][-1].get_attention_weight()


# ------------------------------------------------------------------------------------------------------
# a9-a11: the cross-modal model
# ------------------------------------------------------------------------------------------------------
class W2VVPP(nn.Module):
    """Inference surface of the reference's W2VVPP family (model/model.py:791-1128)."""

    #: operand precision of the similarity GEMM ('fp32' | 'fp16' | 'bf16' | 'fp16x3' | 'bf16x3')
    sim_precision = None

    def _init_vis_net(self, opt):
        raise NotImplementedError

    def _init_txt_net(self, opt):
        raise NotImplementedError

    def __init__(self, opt):
        super().__init__()
        if opt is None:
            return
        self.opt = opt
        self._init_vis_net(opt)
        self._init_txt_net(opt)

    def forward(self, *a, **k):
        raise NotImplementedError('training step (model/model.py:964-1001) is out of scope; use predict()')

    # -- towers ---------------------------------------------------------------------------------------------
    def encode_video(self, vis_input, vis_frame_feat_dict=None):
        """Alias named by BASELINE.json; = self.vis_net(vis_input, vis_frame_feat_dict_input=...)."""
        with torch.no_grad():
            return self.vis_net(vis_input, vis_frame_feat_dict_input=vis_frame_feat_dict if vis_frame_feat_dict is not None else {})

    def encode_text(self, caption_feat_dict):
        with torch.no_grad():
            return self.txt_net(caption_feat_dict)

    # -- similarity -----------------------------------------------------------------------------------------
    @staticmethod
    def compute_sim(query_embs, retro_embs, measure='cosine', device=None):
        if measure != 'cosine':
            raise NotImplementedError("measure '%s' is never configured on the path (base_config.py:92)" % measure)
        dev = device if device is not None else globals()['device']
        return _loss.cosine_sim(query_embs.to(dev), retro_embs.to(dev), W2VVPP.sim_precision)

    def get_txt2vis_matrix(self, txt_embs, vis_embs, measure='cosine', precision=None):
        """2-D: cosine; 3-D: mean over heads of per-head cosine == one GEMM over the concatenated,
        per-head normalised embeddings divided by H."""
        if measure != 'cosine':
            raise NotImplementedError("measure '%s'" % measure)
        if txt_embs.dim() != vis_embs.dim() or txt_embs.dim() not in (2, 3):
            raise ValueError('txt_embs %s / vis_embs %s' % (tuple(txt_embs.shape), tuple(vis_embs.shape)))
        precision = precision or self.sim_precision or _loss.DEFAULT_PRECISION
        heads = txt_embs.shape[1] if txt_embs.dim() == 3 else 1
        T = ops.pack_rows(to_device_and_float16(txt_embs).contiguous(), True, 1e-13, precision)
        V = ops.pack_rows(to_device_and_float16(vis_embs).contiguous(), True, 1e-13, precision)
        return ops.sim_gemm(T, V, heads=heads)

    # -- predict --------------------------------------------------------------------------------------------
    def _embed_videos(self, vis_loader):
        embs, idxs_list, vis_ids = [], [], []
        for output_dict in vis_loader:
            vis_input, idxs, batch_vis_ids = output_dict['vis_feat_dict'], output_dict['idxs'], output_dict['vis_ids']
            frame_dict = output_dict.get('vis_frame_feat_dict', {})
            idxs_list.append(list(idxs))
            embs.append(self.vis_net(vis_input, vis_frame_feat_dict_input=frame_dict))
            vis_ids.extend(batch_vis_ids)
        return torch.cat(embs, dim=0), idxs_list, vis_ids

    #: operand precision of retrieve() / predict() when none is given: the score matrix of the drop-in path goes to the host anyway
    #: (1.6 GB over PCIe at C4 is 40x the GEMM), so it takes the split-product GEMM whose scores are fp32-class (~2e-7) like the
    #: reference's own; bench.py / retrieval.evaluate choose their precision themselves
    predict_precision = 'fp16x3'
    #: retrieve() / predict() run each tower ONCE over all loader batches (False: one launch set per batch, like the reference's loop)
    coalesce_loader_batches = True
    #: ... for collections of at most this many videos / captions: the batches of a loader are concatenated in host memory first (the
    #: reference bounds host memory the same way: predict_batch above 5e4 videos, model/model.py:1081-1128)
    coalesce_max_items = 400000

    def _embed_whole(self, vis_loader, txt_loader):
        """Both towers once over the whole matrices: every FC projection of both towers in ONE grouped launch, one fuse launch per
        side -- instead of one launch set per loader batch (780 of them at C4 with the shipped batch size of 64,
        shell/retrieval_task.sh:161).  Loaders that can hand the matrices over do (`whole()`: laff_amd.data.Bulk*Loader); the batches of
        any other loader -- the reference's own DataLoaders -- are collected and concatenated first (`coalesce_batches`).  Row-wise
        arithmetic, so the embeddings are bit-identical to the per-batch route on tiles of the same kind.  Returns None when the
        batches cannot be concatenated (frame tensors padded to different lengths, foreign value types): per-batch route."""
        if hasattr(vis_loader, 'whole') and hasattr(txt_loader, 'whole'):
            out, (cap, _, txt_ids) = vis_loader.whole(), txt_loader.whole()
        else:
            # the loaders are walked ONCE: what was collected is handed back to the per-batch route when it cannot be concatenated
            # (errors raised by the loaders themselves propagate)
            vb, tb = list(vis_loader), list(txt_loader)
            self._collected_batches = (vb, tb)
            try:
                out = coalesce_batches(vb)
                cap, txt_ids = coalesce_batches([b[0] for b in tb]), [i for b in tb for i in b[2]]
            except (TypeError, ValueError, RuntimeError):
                return None
            if out is None or cap is None:
                return None
            self._collected_batches = None
        pending = []
        fin_v = self.vis_net.prepare(out['vis_feat_dict'], out.get('vis_frame_feat_dict', {}), pending)
        fin_t = self.txt_net.prepare(cap, pending)
        run_fc(pending)
        return fin_v(), [list(out['idxs'])], list(out['vis_ids']), fin_t(), list(txt_ids)

    def retrieve(self, txt_loader, vis_loader, measure='cosine', record_emb=False, precision=None):
        """Device-resident version of predict(): returns (S_device (Nt,Nv) fp32, txt_ids, vis_ids).

        When every caption id names its video the reference's way (`txt_id.split('#')[0]` in vis_ids, predictor.py:241), S is
        produced by the exact-rank pipeline: the text->video ranks counted from it (predictor.t2v_ranks, or an argsort on the host
        as the reference does) are the ranks of the exact cosine scores whatever the operand precision; they are also kept in
        `self.last_t2v_ranks`, and the pipeline's state in `self.last_rank_state` (exact video->text positions:
        predictor.retrieval_metrics(S, txt_ids, vis_ids, state=model.last_rank_state))."""
        if measure != 'cosine':
            raise NotImplementedError("measure '%s'" % measure)
        self.eval()
        if not hasattr(self, 'video_all_embs'):
            self.video_all_embs = None
            self.video_idxs_list = []
        precision = precision or self.sim_precision or self.predict_precision
        with torch.no_grad():
            whole = None
            self._collected_batches = None
            vis_loader_given = vis_loader
            if (self.coalesce_loader_batches and (not record_emb or self.video_all_embs is None) and
                    0 < _loader_len(vis_loader) <= self.coalesce_max_items and 0 < _loader_len(txt_loader) <= self.coalesce_max_items):
                whole = self._embed_whole(vis_loader, txt_loader)
            if whole is None and self._collected_batches is not None:
                vis_loader, txt_loader = self._collected_batches       # walk what was collected, not the loaders a second time
                self._collected_batches = None
            if whole is not None:
                self.video_all_embs, self.video_idxs_list, self.vis_ids, txt_all, txt_ids = whole
            else:
                if not record_emb or self.video_all_embs is None:
                    self.video_all_embs, self.video_idxs_list, self.vis_ids = self._embed_videos(vis_loader)
                txt_ids, txt_embs = [], []
                for caption_feat_dict, txt_idxs, batch_txt_ids in txt_loader:
                    txt_embs.append(self.txt_net(caption_feat_dict))
                    txt_ids.extend(batch_txt_ids)
                txt_all = torch.cat(txt_embs, dim=0)
            cols = np.concatenate([np.asarray(i, dtype=np.int64) for i in self.video_idxs_list])
            vis_used = self.video_all_embs
            identity = np.array_equal(cols, np.arange(len(cols)))
            if not identity:   # the reference indexes the cached embeddings BY dataset index (:1066)
                vis_used = self.video_all_embs[torch.as_tensor(cols, device=self.video_all_embs.device)]
            self.last_t2v_ranks = None
            self.last_rank_state = None          # ops.RankState of the exact-rank pass: predictor.retrieval_metrics(S, ..., state=) ranks V2T exactly with it
            owner = None
            if identity and len(txt_ids) and len(self.vis_ids) == vis_used.shape[0]:
                from ..predictor import gt_columns
                try:
                    owner = gt_columns(txt_ids, self.vis_ids)
                except (IndexError, ValueError, AttributeError):
                    owner = None                 # ids do not follow the protocol: plain scores
            if owner is not None:
                gt = torch.as_tensor(owner, dtype=torch.int32, device=txt_all.device)
                Et, Ev = txt_all.contiguous(), vis_used.contiguous()
                # The pair list of the exact-rank pipeline overflows only on degenerate scores (thousands of videos inside one
                # query's error band).  This path goes to the host anyway, so the flag is read (one small synchronising copy) and
                # the pass repeated: first with an 8x larger list, then with hi/lo split operands (band ~1e-6 instead of ~4e-4).
                # Nothing is published from an overflowed pass.
                attempts = [(precision, None), (precision, 8 * ops.default_pair_cap(Et.shape[0]))]
                if precision not in ('fp16x3', 'bf16x3'):
                    attempts.append(('fp16x3', 8 * ops.default_pair_cap(Et.shape[0])))
                for prec, cap in attempts:
                    T = ops.pack_rows(to_device_and_float16(Et), True, 1e-13, prec)
                    V = ops.pack_rows(to_device_and_float16(Ev), True, 1e-13, prec)
                    S, count, st = ops.exact_ranks(Et, Ev, T, V, gt, pair_cap=cap)
                    if not st.overflowed():
                        break
                    del S, count, st
                else:
                    raise RuntimeError('laff_amd: the pair list of the exact-rank pipeline overflowed even with split operands and an 8x '
                                       'list (%d texts x %d videos): the scores are degenerate' % (Et.shape[0], Ev.shape[0]))
                self.last_t2v_ranks = count + 1
                if identity:
                    self.last_rank_state = st
            else:
                S = self.get_txt2vis_matrix(txt_all, vis_used, measure, precision)
            if not identity:
                full = torch.zeros((S.shape[0], len(vis_loader_given.dataset)), device=S.device, dtype=S.dtype)
                full[:, torch.as_tensor(cols, device=S.device)] = S
                S = full
        return S, txt_ids, self.vis_ids

    def predict(self, txt_loader, vis_loader, measure, record_emb=False):
        """Same contract as the reference (model/model.py:1018-1079): (np.float32[Nt,Nv] in loader row order,
        txt_ids, vis_ids).  Embeddings stay in HBM, all pairs are scored by one GEMM, one D2H copy at the end."""
        S, txt_ids, vis_ids = self.retrieve(txt_loader, vis_loader, measure, record_emb)
        return self._scores_to_host(S), txt_ids, vis_ids

    def _scores_to_host(self, S):
        """The score matrix as a numpy array: one asynchronous copy into a PINNED host buffer -- 1.6 GB at C4 take ~30 ms that way
        against ~200 ms through pageable memory.  The array owns that buffer; the buffer is taken again by the next call only once the
        caller has dropped the array (weak reference), so results never alias."""
        if not S.is_cuda or S.numel() < (1 << 20):
            return S.cpu().numpy()
        import weakref
        buf, ref = getattr(self, '_pinned_scores', None), getattr(self, '_pinned_scores_user', None)
        if buf is None or buf.numel() < S.numel() or (ref is not None and ref() is not None):
            self._pinned_scores = self._pinned_scores_user = None
            try:
                buf = torch.empty((S.numel(),), dtype=S.dtype, pin_memory=True)
            except RuntimeError:
                return S.cpu().numpy()
            self._pinned_scores = buf
        host = buf[:S.numel()].view(S.shape)
        host.copy_(S, non_blocking=True)
        torch.cuda.current_stream(S.device).synchronize()
        out = host.numpy()
        self._pinned_scores_user = weakref.ref(out)
        return out

    def predict_batch(self, txt_loader, vis_loader, measure, record_emb=False):
        """The reference switches to a re-embedding loop above 5e4 videos to bound host memory (:1081-1128);
        with 288 GB of HBM the single-pass path covers it, results are identical."""
        return self.predict(txt_loader, vis_loader, measure, False)


class W2VVPP_MutiVis(W2VVPP):
    def _init_txt_net(self, opt):
        if opt.txt_attention == 'concat':
            raise NotImplementedError("txt_attention 'concat' (MultiScaleTxtNet) is outside the LAFF hot path")
        self.txt_net = MultiScaleTxtEncoderAttention(opt)

    def _init_vis_net(self, opt):
        if opt.vis_attention == 'concat':
            raise NotImplementedError("vis_attention 'concat' (VisTransformNet) is outside the LAFF hot path")
        self.vis_net = VisMutiTransformNetAddAttnetion(opt, opt.vis_fc_layers[0])

    def change_raw_global_emb_weight(self):
        """Linear decay of gw, once per epoch in the reference (model/model.py:1910-1941).  Like the reference it only looks
        for an attribute called `attention_layer`: the FrameLAFF video tower (`vis_attention_layer`) is never decayed."""
        for net, rate in ((self.txt_net, self.opt.txt_attention_global_decay_rate),
                          (self.vis_net, self.opt.vis_attention_global_decay_rate)):
            layer = getattr(net, 'attention_layer', None)
            if layer is not None and hasattr(layer, 'get_raw_global_emb_weight'):
                layer.change_raw_global_emb_weight(max(0.0, rate - 1 + layer.get_raw_global_emb_weight()))


class W2VVPP_MultiHeadAttention(W2VVPP_MutiVis):
    def get_txt2vis_matrix_each_head(self, txt_embs, vis_embs, measure='cosine'):
        return torch.stack([self.get_txt2vis_matrix(txt_embs[:, h, :].contiguous(), vis_embs[:, h, :].contiguous(), measure)
                            for h in range(txt_embs.shape[1])], dim=0)

    def predict_each_head(self, txt_loader, vis_loader, measure):
        """(H, Nt, Nv) per-head score cubes (model/model.py:2058-2098)."""
        self.eval()
        with torch.no_grad():
            vis_all, idxs_list, vis_ids = self._embed_videos(vis_loader)
            txt_ids, txt_embs = [], []
            for caption_feat_dict, txt_idxs, batch_txt_ids in txt_loader:
                txt_embs.append(self.txt_net(caption_feat_dict))
                txt_ids.extend(batch_txt_ids)
            scores = self.get_txt2vis_matrix_each_head(torch.cat(txt_embs, 0), vis_all, measure)
        return scores.cpu().numpy(), txt_ids, vis_ids


class W2VVPP_MutiVisFrameFeat(W2VVPP_MutiVis):
    def _init_vis_net(self, opt):
        self.vis_net = VisMutiTransformNetPlusFrameFeat(opt)


def get_model(name, device_, config):
    """Registry of the reference (model/model.py:2501-2519); keys outside the hot path are refused."""
    global device
    global float16
    device = torch.device(device_)
    float16 = config.float16
    NAME_TO_MODELS = {
        'FrameLAFF': W2VVPP_MutiVisFrameFeat,
        'w2vpp_mutivis_attention': W2VVPP_MutiVis,
        'LAFF': W2VVPP_MultiHeadAttention,
    }
    if name in ('W2VVPP', 'End2EndClip'):
        raise NotImplementedError("model '%s' is outside the LAFF hot path (SURVEY.md section 2)" % name)
    assert name in NAME_TO_MODELS, '%s not supported.' % name
    model_ = NAME_TO_MODELS[name](config)
    model_ = model_.float().to(device)
    return model_
