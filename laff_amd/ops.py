"""Tensor-level wrappers over the C ABI (include/laff_hip.h).

torch is used here for device memory and streams only: every function takes CUDA (ROCm) fp32 tensors,
passes raw pointers to liblaff_hip.so on the current torch stream and returns torch tensors.  There is no
CPU or eager-PyTorch fallback: CPU tensors are rejected.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (ACT, ATT_JUST_AVERAGE, ATT_L2NORM_EACH_HEAD, ATT_MUL, ATT_NO_SPLIT_HEAD, ATT_WITH_AVE, PREC,
                   FcProblem, FcSplitProblem, FcStripProblem, Plane, check, FcFusedProblem, RankSide)

__all__ = ['rank_resolve_metrics', 'rank_prepare', 'rank_prepare_text', 'rank_band_video', 'rank_export_pairs', 'rank_resolve_list', 'sim_gemm_banded', 'rank_resolve', 'exact_ranks', 'RankState', 'topk_rows', 'topk_from_operands', 'alloc_scores', 'frame_fuse_grouped', 'fc_act_bn_fused_grouped', 'fused_split_eligible', 'fc_strip_pack', 'fc_strip_eligible', 'fc_act_bn_strip_grouped', 'StripWeights', 'margin_loss', 'fc_gather_act_bn', 'fc_act_bn', 'fc_act_bn_grouped', 'fc_act_bn_split_grouped', 'split_rows', 'row_dot_gt', 'rank_metrics_async', 'fuse', 'frame_fuse', 'pack_rows', 'sim_gemm', 'gather_gt', 'rank_count', 'v2t_count', 'v2t_count_exact', 'reset_contexts', 'FusedPrepare', 'fused_prepare_eligible',
           'rank_metrics', 'attention_flags', 'PREC', 'default_prescale']

_ctx = {}

#: optional per-launch profiler (bench.py): object with begin(name) / end(name), called on the launching stream
profiler = None


def _call(name, fn, *args):
    if profiler is None:
        check(fn(*args))
    else:
        profiler.begin(name)
        check(fn(*args))
        rep = getattr(profiler, 'repeat', None)       # tools/energy_table.py: the launch issued n times in a row
        if rep is not None:
            for _ in range(rep(name) - 1):
                check(fn(*args))
        profiler.end(name)


def _context(device):
    """One laff_ctx per device ordinal, re-bound to torch's current stream at every call."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    lib = _lib.load()
    if idx not in _ctx:
        h = C.c_void_p()
        check(lib.laff_ctx_create(idx, None, C.byref(h)))
        _ctx[idx] = h
    h = _ctx[idx]
    check(lib.laff_ctx_set_stream(h, C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)))
    return lib, h


def stamp(slots, i):
    """laff_stamp: the device wall clock into slots[i] (int64 device tensor), ordered on the current stream, capturable."""
    lib, h = _context(slots.device)
    check(lib.laff_stamp(h, C.c_void_p(slots.data_ptr() + 8 * int(i))))


def wall_clock_khz(device):
    lib, h = _context(device)
    khz = C.c_int()
    check(lib.laff_wall_clock_khz(h, C.byref(khz)))
    return int(khz.value)


def reset_contexts():
    """Destroys the cached laff_ctx handles; the next call creates fresh ones (the LAFF_* environment knobs are read then)."""
    lib = _lib.load()
    torch.cuda.synchronize()
    for k, h in list(_ctx.items()):
        if isinstance(k, int):                 # (other keys: per-context scratch kept alive for captured graphs)
            lib.laff_ctx_destroy(h)
    _ctx.clear()


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('%s must be a CUDA (ROCm) tensor: laff_amd has no CPU path' % name)
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    return t


def _rows(t, name):
    """2-D view with unit inner stride; returns (tensor, ld)."""
    _dev(t, name)
    if t.dim() != 2 or (t.numel() and t.stride(1) != 1):
        raise ValueError('%s must be 2-D with contiguous rows, got shape %s strides %s' % (name, tuple(t.shape), t.stride()))
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def alloc_scores(Nt, Nv, device):
    """(Nt, Nv) fp32 score matrix whose rows start on 128-byte lines (row pitch = Nv rounded up to 32 floats; a view when Nv is not
    a multiple of 32): the GEMM stores whole lines, which the memory system takes ~25 % faster than rows that straddle them
    (40000 x 10000: 0.496 -> 0.480 ms).  Every kernel of the library takes the row pitch (lds / ldo)."""
    pitch = (Nv + 31) & ~31
    if pitch == Nv or Nv < 1024:
        return torch.empty((Nt, Nv), device=device, dtype=torch.float32)
    return torch.empty((Nt, pitch), device=device, dtype=torch.float32)[:, :Nv]


def attention_flags(with_ave=False, mul=False, l2norm_each_head=False, split_head=True, just_average=False):
    return ((ATT_WITH_AVE if with_ave else 0) | (ATT_MUL if mul else 0) |
            (ATT_L2NORM_EACH_HEAD if l2norm_each_head else 0) | (0 if split_head else ATT_NO_SPLIT_HEAD) |
            (ATT_JUST_AVERAGE if just_average else 0))


def fc_act_bn(x, weight, bias=None, bn_scale=None, bn_shift=None, activation=None, out=None):
    """Y = act(x @ weight.T + bias) * bn_scale + bn_shift   (TransformNet.forward, eval mode)."""
    x, ldx = _rows(x, 'x')
    w, ldw = _rows(weight, 'weight')
    N, Dk = x.shape
    D = w.shape[0]
    if w.shape[1] != Dk:
        raise ValueError('weight is %s but x has %d columns' % (tuple(w.shape), Dk))
    for t, nm in ((bias, 'bias'), (bn_scale, 'bn_scale'), (bn_shift, 'bn_shift')):
        if t is not None:
            _dev(t, nm)
            if t.numel() != D or not t.is_contiguous():
                raise ValueError('%s must be a contiguous vector of %d' % (nm, D))
    if out is None:
        out = torch.empty((N, D), device=x.device, dtype=torch.float32)
    y, ldy = _rows(out, 'out')
    lib, h = _context(x.device)
    _call('fc_act_bn', lib.laff_fc_act_bn, h, _ptr(x), N, Dk, ldx, _ptr(w), ldw, _ptr(bias), _ptr(bn_scale), _ptr(bn_shift), D,
                             ACT[activation], _ptr(y), ldy)
    return out


def fc_act_bn_grouped(problems):
    """Several independent projections in one launch.  problems: list of dicts with keys x, weight and optional
    bias, bn_scale, bn_shift, activation, out.  Returns the list of outputs."""
    if not problems:
        return []
    arr = (FcProblem * len(problems))()
    outs, keep = [], []
    for i, q in enumerate(problems):
        x, ldx = _rows(q['x'], 'x')
        w, ldw = _rows(q['weight'], 'weight')
        N, Dk = x.shape
        D = w.shape[0]
        if w.shape[1] != Dk:
            raise ValueError('problem %d: weight is %s but x has %d columns' % (i, tuple(w.shape), Dk))
        vecs = []
        for nm in ('bias', 'bn_scale', 'bn_shift'):
            t = q.get(nm)
            if t is not None:
                _dev(t, nm)
                if t.numel() != D or not t.is_contiguous():
                    raise ValueError('%s must be a contiguous vector of %d' % (nm, D))
            vecs.append(t)
        out = q.get('out')
        if out is None:
            out = torch.empty((N, D), device=x.device, dtype=torch.float32)
        y, ldy = _rows(out, 'out')
        arr[i] = FcProblem(x.data_ptr(), N, Dk, ldx, w.data_ptr(), ldw, *[t.data_ptr() if t is not None else None for t in vecs],
                           D, ACT[q.get('activation')], y.data_ptr(), ldy)
        outs.append(out)
        keep.append((x, w, vecs))
    lib, h = _context(problems[0]['x'].device)
    _call('fc_act_bn', lib.laff_fc_act_bn_grouped, h, arr, len(problems))
    return outs


def fc_gather_act_bn(x_csr, weight_t, bias=None, bn_scale=None, bn_shift=None, activation=None):
    """TransformNet on a sparse (torch CSR) feature matrix: gather-sum of rows of weight_t = W^T [Dk, D]."""
    if x_csr.layout != torch.sparse_csr or not x_csr.is_cuda:
        raise RuntimeError('x_csr must be a CUDA torch.sparse_csr tensor')
    N, Dk = x_csr.shape
    wt, ldwt = _rows(weight_t, 'weight_t')
    if wt.shape[0] != Dk:
        raise ValueError('weight_t is %s but x has %d columns' % (tuple(wt.shape), Dk))
    D = wt.shape[1]
    crow = x_csr.crow_indices().to(torch.int32).contiguous()
    col = x_csr.col_indices().to(torch.int32).contiguous()
    val = x_csr.values().to(torch.float32).contiguous()
    for t, nm in ((bias, 'bias'), (bn_scale, 'bn_scale'), (bn_shift, 'bn_shift')):
        if t is not None:
            _dev(t, nm)
    out = torch.empty((N, D), device=wt.device, dtype=torch.float32)
    lib, h = _context(wt.device)
    _call('fc_gather', lib.laff_fc_gather_act_bn, h, _ptr(crow), _ptr(col), _ptr(val), N, Dk, _ptr(wt), ldwt, _ptr(bias),
          _ptr(bn_scale), _ptr(bn_shift), D, ACT[activation], _ptr(out), D)
    return out


LOSS_FLAGS = {'max_violation': 1, 'mean': 2, 'i2t': 4, 't2i': 8}


def margin_loss(s, im, margin, max_violation=True, cost_style='sum', direction='t2i', want_grad=True):
    """MarginRankingLoss over heads (laff_margin_loss).  s, im: (B, d) or (B, H, d) fp32 CUDA tensors.
    Returns (loss 0-d tensor, d_s, d_im) -- the gradients are None when want_grad is False."""
    if s.shape != im.shape or s.dim() not in (2, 3):
        raise ValueError('s and im must both be (B, d) or (B, H, d); got %s and %s' % (tuple(s.shape), tuple(im.shape)))
    if direction not in ('i2t', 't2i', 'bidir'):
        raise ValueError("direction must be 'i2t', 't2i' or 'bidir'")
    if cost_style not in ('sum', 'mean'):
        raise ValueError("cost_style must be 'sum' or 'mean'")
    s_c = _dev(s.contiguous(), 's')
    im_c = _dev(im.contiguous(), 'im')
    B, H, d = (s.shape[0], 1, s.shape[1]) if s.dim() == 2 else s.shape
    flags = (LOSS_FLAGS['max_violation'] if max_violation else 0) | (LOSS_FLAGS['mean'] if cost_style == 'mean' else 0)
    flags |= {'i2t': LOSS_FLAGS['i2t'], 't2i': LOSS_FLAGS['t2i'], 'bidir': LOSS_FLAGS['i2t'] | LOSS_FLAGS['t2i']}[direction]
    lib, h = _context(s_c.device)
    nbytes = C.c_size_t()
    rc = lib.laff_margin_loss_workspace_bytes(B, H, d, C.byref(nbytes))
    if rc:
        raise RuntimeError(lib.laff_last_error().decode())
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=s_c.device)
    loss = torch.empty((), dtype=torch.float32, device=s_c.device)
    d_s = torch.empty_like(s_c) if want_grad else None
    d_im = torch.empty_like(im_c) if want_grad else None
    _call('margin_loss', lib.laff_margin_loss, h, _ptr(s_c), _ptr(im_c), B, H, d, float(margin), flags, _ptr(loss), _ptr(d_s),
          _ptr(d_im), _ptr(ws), nbytes.value)
    return loss, d_s, d_im


class SplitOperand:
    """fp16 hi/lo split of an fp32 matrix (laff_split_rows): buffer [2][N][Kp] + per-row reciprocal scales."""

    def __init__(self, buf, rscale, N, K):
        self.buf, self.rscale, self.N, self.K = buf, rscale, N, K


def split_rows(x):
    x, ldx = _rows(x, 'x')
    N, K = x.shape
    lib, h = _context(x.device)
    nbytes = C.c_size_t()
    check(lib.laff_split_rows_bytes(N, K, C.byref(nbytes)))
    buf = torch.empty((max(nbytes.value, 16),), device=x.device, dtype=torch.uint8)
    rscale = torch.empty((max(N, 1),), device=x.device, dtype=torch.float32)
    _call('split_rows', lib.laff_split_rows, h, _ptr(x), N, K, ldx, _ptr(buf), _ptr(rscale))
    return SplitOperand(buf, rscale, N, K)


def split_rows_grouped(xs):
    """split_rows for several matrices in one launch."""
    if not xs:
        return []
    n = len(xs)
    lib, h = _context(xs[0].device)
    X, N, K, LD, O, R = (C.c_void_p * n)(), (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    outs = []
    for i, x in enumerate(xs):
        x, ldx = _rows(x, 'x')
        nbytes = C.c_size_t()
        check(lib.laff_split_rows_bytes(x.shape[0], x.shape[1], C.byref(nbytes)))
        buf = torch.empty((max(nbytes.value, 16),), device=x.device, dtype=torch.uint8)
        rscale = torch.empty((max(x.shape[0], 1),), device=x.device, dtype=torch.float32)
        X[i], N[i], K[i], LD[i], O[i], R[i] = x.data_ptr(), x.shape[0], x.shape[1], ldx, buf.data_ptr(), rscale.data_ptr()
        outs.append(SplitOperand(buf, rscale, x.shape[0], x.shape[1]))
    _call('split_rows', lib.laff_split_rows_grouped, h, n, X, N, K, LD, O, R)
    return outs


def fc_act_bn_split_grouped(problems):
    """fc_act_bn_grouped on the fp16 matrix pipe with fp32-class accuracy.  problems: dicts with x (fp32 tensor or
    SplitOperand), weight_split (SplitOperand of W), optional bias / bn_scale / bn_shift / activation / out."""
    if not problems:
        return []
    arr = (FcSplitProblem * len(problems))()
    outs, keep = [], []
    dev = problems[0]['weight_split'].buf.device
    todo = [i for i, q in enumerate(problems) if not isinstance(q['x'], SplitOperand)]
    split = dict(zip(todo, split_rows_grouped([problems[i]['x'] for i in todo])))      # all inputs in ONE launch
    for i, q in enumerate(problems):
        xs = split.get(i, q['x'])
        ws = q['weight_split']
        if xs.K != ws.K:
            raise ValueError('problem %d: x has %d columns, weight %d' % (i, xs.K, ws.K))
        N, D = xs.N, ws.N
        vecs = []
        for nm in ('bias', 'bn_scale', 'bn_shift'):
            t = q.get(nm)
            if t is not None:
                _dev(t, nm)
                if t.numel() != D or not t.is_contiguous():
                    raise ValueError('%s must be a contiguous vector of %d' % (nm, D))
            vecs.append(t)
        out = q.get('out')
        if out is None:
            out = torch.empty((N, D), device=dev, dtype=torch.float32)
        y, ldy = _rows(out, 'out')
        arr[i] = FcSplitProblem(xs.buf.data_ptr(), xs.rscale.data_ptr(), N, xs.K, ws.buf.data_ptr(), ws.rscale.data_ptr(),
                                *[t.data_ptr() if t is not None else None for t in vecs], D, ACT[q.get('activation')],
                                y.data_ptr(), ldy)
        outs.append(out)
        keep.append((xs, ws, vecs))
    lib, h = _context(dev)
    _call('fc_act_bn', lib.laff_fc_act_bn_split_grouped, h, arr, len(problems))
    return outs


def fused_split_eligible(x, weight_split):
    """Can laff_fc_act_bn_fused_grouped take this input?  (fp32 CUDA matrix, 16-byte aligned rows, K a multiple of 32.)"""
    return (torch.is_tensor(x) and x.is_cuda and x.layout == torch.strided and x.dtype == torch.float32 and x.dim() == 2 and
            x.shape[1] % 32 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and
            x.shape[1] == weight_split.K)


def fc_act_bn_fused_grouped(problems):
    """fc_act_bn_split_grouped without materialising the split of the inputs: one pass over the inputs for their per-row scales
    (laff_row_scales_grouped), then the GEMM splits them on the way into LDS.  problems: dicts with x (fp32 tensor),
    weight_split (SplitOperand of W), optional bias / bn_scale / bn_shift / activation / out."""
    if not problems:
        return []
    n = len(problems)
    dev = problems[0]['weight_split'].buf.device
    lib, h = _context(dev)
    X, N, K, LD, R = (C.c_void_p * n)(), (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)(), (C.c_void_p * n)()
    arr = (FcFusedProblem * n)()
    outs, keep = [], []
    total = sum(q['x'].shape[0] for q in problems)
    scales = torch.empty((max(total, 1),), device=dev, dtype=torch.float32)      # one buffer, one slice per problem
    at = 0
    for i, q in enumerate(problems):
        x, ldx = _rows(q['x'], 'x')
        ws = q['weight_split']
        if not fused_split_eligible(x, ws):
            raise ValueError('problem %d is not eligible for the fused split (see fused_split_eligible)' % i)
        rs = scales[at:at + x.shape[0]]
        at += x.shape[0]
        X[i], N[i], K[i], LD[i], R[i] = x.data_ptr(), x.shape[0], x.shape[1], ldx, rs.data_ptr()
        D = ws.N
        vecs = []
        for nm in ('bias', 'bn_scale', 'bn_shift'):
            t = q.get(nm)
            if t is not None:
                _dev(t, nm)
                if t.numel() != D or not t.is_contiguous():
                    raise ValueError('%s must be a contiguous vector of %d' % (nm, D))
            vecs.append(t)
        out = q.get('out')
        if out is None:
            out = torch.empty((x.shape[0], D), device=dev, dtype=torch.float32)
        y, ldy = _rows(out, 'out')
        arr[i] = FcFusedProblem(x.data_ptr(), ldx, rs.data_ptr(), x.shape[0], x.shape[1], ws.buf.data_ptr(), ws.rscale.data_ptr(),
                                *[t.data_ptr() if t is not None else None for t in vecs], D, ACT[q.get('activation')],
                                y.data_ptr(), ldy)
        outs.append(out)
        keep.append((x, ws, vecs))
    _call('row_scales', lib.laff_row_scales_grouped, h, n, X, N, K, LD, R)
    _call('fc_act_bn', lib.laff_fc_act_bn_fused_grouped, h, arr, n)
    return outs


class StripWeights:
    """A TransformNet's parameters packed for the strip-form FC (laff_fc_strip_pack): the W stream's LDS image + the per-column
    epilogue constants (bias, activation and folded BatchNorm combined)."""

    def __init__(self, img, D, K, act):
        self.img, self.D, self.K, self.act = img, D, K, act


def fc_strip_pack(weight, bias=None, bn_scale=None, bn_shift=None, activation=None):
    """Pack W [D, 512] (+ bias / folded BatchNorm / activation) once per model for fc_act_bn_strip_grouped."""
    w, ldw = _rows(weight, 'weight')
    D, K = w.shape
    lib, h = _context(w.device)
    nbytes = C.c_size_t()
    check(lib.laff_fc_strip_pack_bytes(D, K, C.byref(nbytes)))
    for t, nm in ((bias, 'bias'), (bn_scale, 'bn_scale'), (bn_shift, 'bn_shift')):
        if t is not None:
            _dev(t, nm)
            if t.numel() != D or not t.is_contiguous():
                raise ValueError('%s must be a contiguous vector of %d' % (nm, D))
    img = torch.empty((nbytes.value,), device=w.device, dtype=torch.uint8)
    _call('fc_strip_pack', lib.laff_fc_strip_pack, h, _ptr(w), ldw, _ptr(bias), _ptr(bn_scale), _ptr(bn_shift), D, K, ACT[activation],
          _ptr(img))
    return StripWeights(img, D, K, ACT[activation])


def fc_strip_eligible(x, D):
    """Can laff_fc_act_bn_strip_grouped take this input?  (fp32 CUDA matrix of 512 columns, 16-byte aligned rows, D % 32 == 0.)"""
    return (torch.is_tensor(x) and x.is_cuda and x.layout == torch.strided and x.dtype == torch.float32 and x.dim() == 2 and
            x.shape[1] == 512 and (x.shape[0] == 0 or (x.stride(1) == 1 and (x.stride(0) % 4 == 0 or x.shape[0] == 1) and
                                                       x.data_ptr() % 16 == 0)) and D % 32 == 0 and D >= 32)


def fc_act_bn_strip_grouped(problems):
    """TransformNet.forward for several features in the strip form (X stationary in registers, no row-scale pass).
    problems: dicts with x (fp32 tensor [N, 512]), strip (StripWeights), optional out.  Returns the list of outputs."""
    if not problems:
        return []
    n = len(problems)
    arr = (FcStripProblem * n)()
    outs, keep = [], []
    dev = problems[0]['strip'].img.device
    for i, q in enumerate(problems):
        x, ldx = _rows(q['x'], 'x')
        sw = q['strip']
        if not fc_strip_eligible(x, sw.D):
            raise ValueError('problem %d is not eligible for the strip form (see fc_strip_eligible)' % i)
        out = q.get('out')
        if out is None:
            out = torch.empty((x.shape[0], sw.D), device=dev, dtype=torch.float32)
        y, ldy = _rows(out, 'out')
        arr[i] = FcStripProblem(x.data_ptr(), max(ldx, 512), x.shape[0], sw.img.data_ptr(), sw.D, sw.act, y.data_ptr(), ldy)
        outs.append(out)
        keep.append((x, sw))
    lib, h = _context(dev)
    _call('fc_act_bn', lib.laff_fc_act_bn_strip_grouped, h, arr, n)
    return outs


def fuse(planes, H, d, w, b, gw, flags, return_weights=False, packed_precision=None, l2norm_planes=False, rank_side=None):
    """planes: list of (src[N, ld-view], tile, scale, shift[, act]) -- act ('tanh' | 'relu' | 'sigmoid' | None) is applied to src
    before the affine (a projection that left its activation + BatchNorm to this kernel).  Returns E (N, H, d) [and softmax
    weights (N, H, L)].
    packed_precision ('fp16' | 'bf16'): also emit the similarity operand in the same launch; returned last.
    l2norm_planes: every plane row is first divided by its l2 norm over all H*d columns (`l2norm(local_embs, dim=2)` of the
    expert-embedding branch, model/model.py:1866-1873): one extra launch computes the norms (laff_plane_row_norms)."""
    L = len(planes)
    arr = (Plane * L)()
    first = planes[0]
    N = (first[5][0] if (len(first) > 5 and first[5] is not None) else first[0]).shape[0]
    keep = []
    for i, pl in enumerate(planes):
        src, tile, scale, shift = pl[:4]
        act = pl[4] if len(pl) > 4 else None
        gather = pl[5] if len(pl) > 5 else None
        for t in (scale, shift):
            if t is not None:
                _dev(t, 'plane affine')
        sp, tp = (scale.data_ptr() if scale is not None else None), (shift.data_ptr() if shift is not None else None)
        if gather is not None:
            # (csr, weight_t, bias): a sparse feature through its FC, gathered inside the fuse launch
            csr, wt, bias = gather
            if csr.layout != torch.sparse_csr or csr.shape[0] != N:
                raise ValueError('gather plane %d must be a CSR matrix with %d rows' % (i, N))
            wt, ldwt = _rows(wt, 'weight_t')
            if wt.shape[0] != csr.shape[1] or wt.shape[1] != H * d or tile or (flags & ATT_NO_SPLIT_HEAD) or d > 512:
                raise ValueError('gather plane %d: weight_t must be (%d, %d), split heads of d <= 512' % (i, csr.shape[1], H * d))
            crow = csr.crow_indices().to(torch.int32).contiguous()
            col = csr.col_indices().to(torch.int32).contiguous()
            val = csr.values().to(torch.float32).contiguous()
            if bias is not None:
                _dev(bias, 'bias')
            arr[i] = Plane(None, 0, 0, sp, tp, ACT[act], crow.data_ptr(), col.data_ptr(), val.data_ptr(), wt.data_ptr(), ldwt,
                           csr.shape[1], bias.data_ptr() if bias is not None else None)
            keep.append((crow, col, val, wt, bias, scale, shift))
            continue
        src, ld = _rows(src, 'plane %d' % i)
        if src.shape[0] != N:
            raise ValueError('plane %d has %d rows, expected %d' % (i, src.shape[0], N))
        need = d if (tile or (flags & ATT_NO_SPLIT_HEAD)) else H * d
        if src.shape[1] != need:
            raise ValueError('plane %d has %d columns, expected %d' % (i, src.shape[1], need))
        arr[i] = Plane(src.data_ptr(), ld, 1 if tile else 0, sp, tp, ACT[act], None, None, None, None, 0, 0, None)
        keep.append((src, scale, shift))
    dev = (first[5][1] if (len(first) > 5 and first[5] is not None) else first[0]).device
    E = torch.empty((N, H, d), device=dev, dtype=torch.float32)
    aw = torch.empty((N, H, L), device=dev, dtype=torch.float32) if return_weights else None
    for t, nm in ((w, 'w'), (b, 'b'), (gw, 'gw')):
        if t is not None:
            _dev(t, nm)
    lib, h = _context(dev)
    if l2norm_planes:
        norms = torch.empty((L, max(N, 1)), device=dev, dtype=torch.float32)
        _call('plane_row_norms', lib.laff_plane_row_norms, h, arr, L, N, H, d, flags, _ptr(norms))
        for i in range(L):
            arr[i].row_scale = norms[i].data_ptr()
        keep.append(norms)
    packed = None
    if packed_precision is not None:
        prescale = default_prescale(packed_precision)
        buf = torch.empty((max(N * H * d * 2, 16),), device=dev, dtype=torch.uint8)
        rs = None
        if rank_side is not None:
            # FusedPrepare (below) for this side: laff_rank_prepare's work for these rows rides in this launch
            rs = rank_side.side_struct(N, E)
        _call('fuse', lib.laff_fuse_packed_rank, h, arr, L, N, H, d, _ptr(w), _ptr(b), _ptr(gw), flags, _ptr(E), _ptr(aw), _ptr(buf),
              PREC[packed_precision], prescale, rs)
        packed = Packed(buf, N, H * d, packed_precision, prescale)
        if rank_side is not None:
            rank_side.done(E, packed)
    else:
        _call('fuse', lib.laff_fuse, h, arr, L, N, H, d, _ptr(w), _ptr(b), _ptr(gw), flags, _ptr(E), _ptr(aw))
    out = (E, aw) if return_weights else (E,)
    if packed_precision is not None:
        out = out + (packed,)
    return out if len(out) > 1 else out[0]


def frame_fuse(frames, lens, w, b, gw, flags):
    """frames (B, Fmax, d) zero padded, lens int32 (B,) or None -> (B, d)."""
    _dev(frames, 'frames')
    if frames.dim() != 3 or not frames.is_contiguous():
        raise ValueError('frames must be contiguous (B, Fmax, d)')
    B, Fmax, d = frames.shape
    if lens is not None:
        _dev(lens, 'lens', torch.int32)
        if lens.numel() != B:
            raise ValueError('lens must have %d entries' % B)
    V = torch.empty((B, d), device=frames.device, dtype=torch.float32)
    lib, h = _context(frames.device)
    _call('frame_fuse', lib.laff_frame_fuse, h, _ptr(frames), _ptr(lens), B, Fmax, d, _ptr(_dev(w, 'w')), _ptr(_dev(b, 'b')),
                              _ptr(gw), flags, _ptr(V))
    return V


def frame_fuse_grouped(frames_list, lens, params, flags, mask=None):
    """frame_fuse for several frame features of the same shape in one launch.  frames_list: [(B, Fmax, d)], params: [(w, b, gw)].
    mask (fp32 (B, >= Fmax) device tensor, rows of ones then zeros: the reference's mask_tensor) replaces lens: the launch sums it."""
    n = len(frames_list)
    B, Fmax, d = frames_list[0].shape
    F, W, Bb, G, Vv = ((C.c_void_p * n)() for _ in range(5))
    outs, keep = [], []
    for i, (fr, (w, b, gw)) in enumerate(zip(frames_list, params)):
        _dev(fr, 'frames')
        if tuple(fr.shape) != (B, Fmax, d) or not fr.is_contiguous():
            raise ValueError('grouped frame features must share one contiguous (B, Fmax, d) shape')
        V = torch.empty((B, d), device=fr.device, dtype=torch.float32)
        F[i], W[i], Bb[i], G[i], Vv[i] = fr.data_ptr(), _dev(w, 'w').data_ptr(), _dev(b, 'b').data_ptr(), _ptr(gw), V.data_ptr()
        outs.append(V)
        keep.append((fr, w, b, gw))
    lib, h = _context(frames_list[0].device)
    if mask is not None:
        _dev(mask, 'mask')
        if mask.dim() != 2 or mask.shape[0] != B or mask.shape[1] < Fmax or mask.stride(1) != 1:
            raise ValueError('mask must be (%d, >= %d) with unit column stride' % (B, Fmax))
        _call('frame_fuse', lib.laff_frame_fuse_grouped_mask, h, n, F, _ptr(mask), mask.stride(0), B, Fmax, d, W, Bb, G, flags, Vv)
        return outs
    if lens is not None:
        _dev(lens, 'lens', torch.int32)
    _call('frame_fuse', lib.laff_frame_fuse_grouped, h, n, F, _ptr(lens), B, Fmax, d, W, Bb, G, flags, Vv)
    return outs


def default_prescale(precision):
    """fp16x3 operands are pre-scaled by 64 (exact) so that the low part of the hi/lo split stays normal.  Single-pass fp16 operands
    are not scaled: the fp16 MFMA honours denormal inputs exactly (tools/debug/denorm_probe.py), an element below 2^-14 of a unit-norm
    row then carries an absolute error of 2^-25 -- that of its normal neighbours -- and the measured band (laff_rank_prepare) covers
    it; with scale == 1 the GEMM epilogue has no product per score to make (sim_strip.hip, SCALE1)."""
    return 64.0 if precision == 'fp16x3' else 1.0


class Packed:
    """GEMM operand produced by pack_rows: raw 16/32-bit buffer + its logical shape and precision."""

    def __init__(self, buf, N, K, precision, prescale):
        self.buf, self.N, self.K, self.precision, self.prescale = buf, N, K, precision, prescale

    def rows(self, a, b):
        """The operand of rows [a, b): a view for one-plane formats, a copy of the two plane slices for the hi/lo split formats
        (their planes are N rows apart)."""
        a, b = max(0, int(a)), min(self.N, int(b))
        esz = 4 if self.precision == 'fp32' else 2
        rb = self.K * esz
        if self.precision in ('fp16x3', 'bf16x3'):
            plane = self.N * rb
            buf = torch.cat([self.buf[a * rb:b * rb], self.buf[plane + a * rb:plane + b * rb]])
        else:
            buf = self.buf[a * rb:b * rb]
        if buf.data_ptr() % 16:
            buf = buf.clone()
        return Packed(buf, b - a, self.K, self.precision, self.prescale)


def pack_rows(E, normalize=True, eps=1e-13, precision='fp16', prescale=None):
    """E (N, H, d) or (N, d): per-(row, head) l2norm (loss.l2norm) then conversion to the GEMM operand format."""
    _dev(E, 'E')
    if E.dim() == 2:
        E = E.unsqueeze(1)
    if E.dim() != 3 or E.stride(2) != 1 or E.stride(1) != E.shape[2]:
        raise ValueError('E must be (N, H, d) with contiguous heads')
    N, H, d = E.shape
    lde = E.stride(0) if N > 1 else H * d
    if prescale is None:
        prescale = default_prescale(precision)
    lib, h = _context(E.device)
    nbytes = C.c_size_t()
    check(lib.laff_packed_bytes(N, H * d, PREC[precision], C.byref(nbytes)))
    buf = torch.empty((max(nbytes.value, 16),), device=E.device, dtype=torch.uint8)
    _call('pack_rows', lib.laff_pack_rows, h, _ptr(E), N, H, d, lde, 1 if normalize else 0, eps, prescale, PREC[precision], _ptr(buf))
    return Packed(buf, N, H * d, precision, prescale)


def sim_gemm(T, V, heads=1, out=None, want_scores=True, gt_col=None, s_gt=None, count=None, col0=0):
    """S = T.V^T / (heads * prescale^2) on Packed operands; optional fused ground-truth rank count."""
    if T.precision != V.precision or T.K != V.K:
        raise ValueError('operands differ in precision or K')
    dev = T.buf.device
    S = None
    lds = V.N
    if want_scores:
        S = out if out is not None else alloc_scores(T.N, V.N, dev)
        S, lds = _rows(S, 'out')
        if tuple(S.shape) != (T.N, V.N):
            raise ValueError('out must be (%d, %d)' % (T.N, V.N))
    if gt_col is not None:
        _dev(gt_col, 'gt_col', torch.int32)
        _dev(s_gt, 's_gt')
        _dev(count, 'count', torch.int32)
    scale = 1.0 / (heads * T.prescale * V.prescale)
    lib, h = _context(dev)
    _call('sim_gemm', lib.laff_sim_gemm, h, _ptr(T.buf), _ptr(V.buf), T.N, V.N, T.K, scale, PREC[T.precision], _ptr(S), lds,
                            _ptr(gt_col), col0, _ptr(s_gt), _ptr(count))
    return S


def row_dot_gt(T, V, gt_col, heads=1, col0=0, zero_count=None):
    """s_gt[t] = <T[t], V[gt[t]-col0]> / (heads * prescale^2) on the packed operands; -inf outside this shard.
    zero_count: optional int32 (Nt,) tensor cleared by the same launch (the accumulator of the fused count)."""
    _dev(gt_col, 'gt_col', torch.int32)
    if zero_count is not None:
        _dev(zero_count, 'zero_count', torch.int32)
        if zero_count.numel() != T.N or not zero_count.is_contiguous():
            raise ValueError('zero_count must be a contiguous int32 vector of %d' % T.N)
    out = torch.empty((T.N,), device=T.buf.device, dtype=torch.float32)
    scale = 1.0 / (heads * T.prescale * V.prescale)
    lib, h = _context(T.buf.device)
    _call('row_dot_gt', lib.laff_row_dot_gt, h, _ptr(T.buf), _ptr(V.buf), T.N, V.N, T.K, scale, PREC[T.precision], _ptr(gt_col),
          col0, _ptr(out), _ptr(zero_count))
    return out


class RankState:
    """What the exact-rank pipeline carries between its three launches (laff_rank_prepare -> laff_sim_gemm_banded ->
    laff_rank_resolve): the exact ground-truth scores (fp64; all-reduce MAX them when videos are sharded), the two band vectors,
    the count accumulator and the list of pairs inside the band."""

    def __init__(self, Et, Ev, T, V, heads, gt_col, col0, s_gt64, band_t, band_v, count, pairs, pair_cap):
        self.Et, self.Ev, self.T, self.V, self.heads, self.gt_col, self.col0 = Et, Ev, T, V, heads, gt_col, col0
        self.s_gt64, self.band_t, self.band_v, self.count, self.pairs, self.pair_cap = s_gt64, band_t, band_v, count, pairs, pair_cap

    GROUP_WORDS, GROUP_CHUNK = 24, 64      # the strip kernel's list (sim_strip.hip): entries of 24 words in chunks of 64

    def _header(self):
        h = self.pairs[:4].cpu().tolist()
        return h[0] & 0xffffffff, h[1], h[2] & 0xffffffff, h[3] & 0xffffffff

    def listed_pairs(self):
        """(number of pairs inside the error band that the GEMM handed to laff_rank_resolve, overflow flag) -- synchronises;
        diagnostics only.  Two list layouts (the header's third word tells them apart):
          tiled kernel: header {overflow count, overflow flag, A, chunk}, A slots of per-wavefront segments (unused slots have row
                        0xffffffff), then the overflow pairs;
          strip kernel: header {chunks taken, overflow flag, NW | 1 << 31, NCH}, NCH per-chunk entry counts, then NCH chunks of 64
                        entries {row, colbase, lo, hi | mask16, the row's ground-truth column, 0, 0 | 16 raw accumulators}: the
                        pairs are the listed elements with lo <= x <= hi (the test laff_rank_resolve applies)."""
        return int(self.pair_indices().shape[0]), self.overflowed()

    def overflowed(self):
        """The list's overflow flag (one 16-byte copy; synchronises): pairs were dropped, the counts are poisoned."""
        return bool(self._header()[1])

    def pair_indices(self):
        """(n, 2) int64 tensor of the (row, col) pairs inside the band -- synchronises; diagnostics / tests only."""
        n_over, _, third, fourth = self._header()
        if not (third & 0x80000000):
            reg_a = third
            seg = self.pairs[4:4 + 2 * reg_a].view(-1, 2)
            over = self.pairs[4 + 2 * reg_a:4 + 2 * (reg_a + min(n_over, max(self.pair_cap - reg_a, 0)))].view(-1, 2)
            return torch.cat([seg[seg[:, 0] != -1], over]).long()
        nw, nch, W, CH = third & 0x7fffffff, fourth, self.GROUP_WORDS, self.GROUP_CHUNK
        cnt_words = (nch + 3) & ~3
        nchunks = min(nw + n_over, nch)                   # (header word 0: chunks taken from the pool)
        counts = self.pairs[4:4 + nchunks].long().clamp(max=CH)
        ent = self.pairs[4 + cnt_words:4 + cnt_words + nchunks * CH * W].view(nchunks, CH, W)
        live = torch.arange(CH, device=ent.device)[None, :] < counts[:, None]
        ent = ent[live]
        if ent.shape[0] == 0:
            return torch.zeros((0, 2), dtype=torch.int64, device=ent.device)
        e = torch.arange(16, device=ent.device)
        x = ent[:, 8:24].view(torch.float32)
        lo, hi = ent[:, 2:3].view(torch.float32), ent[:, 3:4].view(torch.float32)
        rows = ent[:, 0:1].long().expand(-1, 16)
        cols = ent[:, 1:2].long() + 8 * (e >> 2)[None, :] + (e & 3)[None, :]
        listed = ((ent[:, 4:5] >> e[None, :]) & 1).bool() & (cols != ent[:, 5:6].long()) & (x >= lo) & (x <= hi)
        return torch.stack([rows[listed], cols[listed]], dim=1)


def _emb3(E, name):
    _dev(E, name)
    if E.dim() == 2:
        E = E.unsqueeze(1)
    if E.dim() != 3 or not E.is_contiguous():
        raise ValueError('%s must be a contiguous (N, H, d) or (N, d) fp32 tensor' % name)
    return E


def default_pair_cap(Nt):
    """Slots (8 bytes each) of the in-band list.  The strip kernel lists groups of 16 accumulators (96 bytes per dumped lane): 128 Nt
    slots hold ~10 dumped lanes per text -- C4 uses 3.3 (1.3e5 of 4.3e5) with fp16 operands."""
    return max(1 << 20, 128 * int(Nt))


def fused_prepare_eligible(Nt, Nv, H, d, precision):
    """laff_fuse_packed_rank covers split heads of d <= 512 with a single-plane 16-bit operand; the text launch finishes the videos'
    64-column block maxima (it needs ceil(Nv / 64) workgroups of 4 (row, head) items)."""
    return H >= 1 and d <= 512 and d % 4 == 0 and precision in ('fp16', 'bf16') and Nv >= 1 and 16 * Nt * H >= Nv * 4 and 16 * Nt >= Nv


class FusedPrepare:
    """laff_rank_prepare's outputs produced by the two fuse launches of a pass (laff_fuse_packed_rank, videos first): allocate with the
    problem's sizes, hand `.video` / `.text` to the fuse calls of the two towers as `rank_side`, then `.state()` is the RankState
    rank_prepare would have returned."""

    class _Side:
        def __init__(self, owner, side):
            self.owner, self.side = owner, side

        def side_struct(self, N, E):
            o = self.owner
            H = E.shape[1]
            if H != o.heads:
                raise ValueError('the tower produced %d heads, FusedPrepare was made for %d' % (H, o.heads))
            part = tick = None
            if H > 1:
                part = torch.empty((N, H, 2), device=E.device, dtype=torch.float64)
                tick = torch.empty((N,), device=E.device, dtype=torch.int32)
                o._scratch.append((part, tick))
            pp, tp = (part.data_ptr() if part is not None else None), (tick.data_ptr() if tick is not None else None)
            if self.side == 2:
                if N != o.Nv:
                    raise ValueError('the video tower produced %d rows, FusedPrepare was made for %d' % (N, o.Nv))
                return RankSide(2, None, 0, None, 0, None, o.band_v.data_ptr(), None, None, None, pp, tp)
            if N != o.Nt or o.Ev is None:
                raise ValueError('the text tower must run behind the video tower (%d rows, FusedPrepare made for %d)' % (N, o.Nt))
            return RankSide(1, o.gt_col.data_ptr(), o.col0, o.Ev.data_ptr(), o.Nv, o.s_gt64.data_ptr(), o.band_t.data_ptr(),
                            o.band_v.data_ptr(), o.count.data_ptr(), o.pairs.data_ptr(), pp, tp)

        def done(self, E, packed):
            if self.side == 2:
                self.owner.Ev, self.owner.V = E, packed
            else:
                self.owner.Et, self.owner.T = E, packed

    def __init__(self, Nt, Nv, gt_col, col0=0, pair_cap=None, heads=1):
        _dev(gt_col, 'gt_col', torch.int32)
        if gt_col.numel() != Nt or not gt_col.is_contiguous():
            raise ValueError('gt_col must be a contiguous int32 vector of %d' % Nt)
        if gt_col.data_ptr() % 16:
            gt_col = gt_col.clone()
        dev = gt_col.device
        cap = (int(pair_cap) if pair_cap is not None else default_pair_cap(Nt)) & ~3
        if cap < 4:
            raise ValueError('pair_cap must be >= 4')
        self.Nt, self.Nv, self.gt_col, self.col0, self.pair_cap, self.heads = Nt, Nv, gt_col, int(col0), cap, int(heads)
        self.s_gt64 = torch.empty((Nt + 2,), device=dev, dtype=torch.float64)[:Nt]
        self.band_t = torch.empty((Nt + 4,), device=dev, dtype=torch.float32)
        self.band_v = torch.empty((((Nv + 3) & ~3) + (Nv + 63) // 64 + 4,), device=dev, dtype=torch.float32)
        self.count = torch.empty((Nt,), device=dev, dtype=torch.int32)
        self.pairs = torch.empty((4 + 2 * cap,), device=dev, dtype=torch.int32)
        self.Et = self.Ev = self.T = self.V = None
        self._scratch = []
        self.video, self.text = self._Side(self, 2), self._Side(self, 1)

    def state(self):
        if self.Et is None or self.Ev is None:
            raise RuntimeError('FusedPrepare.state(): both fuse launches must have run (videos first)')
        return RankState(self.Et, self.Ev, self.T, self.V, self.heads, self.gt_col, self.col0, self.s_gt64, self.band_t, self.band_v,
                         self.count, self.pairs, self.pair_cap)


def rank_prepare(Et, Ev, T, V, gt_col, col0=0, pair_cap=None, emit_precision=None):
    """First launch of the exact-rank pipeline.  Et (Nt, H, d) / Ev (Nv, H, d): the fp32 embeddings; T / V: the Packed GEMM
    operands made from them; gt_col int32 (Nt,) ground-truth column of every text (global index; col0 = first column of this
    video shard).  Returns a RankState.
    T or V None (with emit_precision 'fp16' | 'bf16', or the other operand's): that operand is PRODUCED by the launch from the
    embedding rows -- E * prescale converted without re-normalising, what pack_rows(E, normalize=False) returns (laff_rank_prepare_emit);
    it is in the returned state."""
    Et, Ev = _emb3(Et, 'Et'), _emb3(Ev, 'Ev')
    Nt, H, d = Et.shape
    Nv = Ev.shape[0]
    emit = (1 if T is None else 0) | (2 if V is None else 0)
    if emit:
        have = T if T is not None else V
        prec = emit_precision or (have.precision if have is not None else None)
        if prec not in ('fp16', 'bf16'):
            raise ValueError("rank_prepare can only produce single-plane 16-bit operands ('fp16' | 'bf16'), got %r" % prec)
        ps = have.prescale if have is not None else default_prescale(prec)
        if T is None:
            T = Packed(torch.empty((max(Nt * H * d * 2, 16),), device=Et.device, dtype=torch.uint8), Nt, H * d, prec, ps)
        if V is None:
            V = Packed(torch.empty((max(Nv * H * d * 2, 16),), device=Et.device, dtype=torch.uint8), Nv, H * d, prec, ps)
    if tuple(Ev.shape[1:]) != (H, d) or T.N != Nt or V.N != Nv or T.K != H * d or V.K != H * d:
        raise ValueError('embeddings %s / %s do not match the operands (%d x %d, %d x %d)' % (tuple(Et.shape), tuple(Ev.shape), T.N, T.K, V.N, V.K))
    if T.precision != V.precision or T.prescale != V.prescale:
        raise ValueError('operands differ in precision or prescale')
    _dev(gt_col, 'gt_col', torch.int32)
    if gt_col.numel() != Nt or not gt_col.is_contiguous():
        raise ValueError('gt_col must be a contiguous int32 vector of %d' % Nt)
    if gt_col.data_ptr() % 16:          # the GEMM fetches gt_col in 16-byte groups
        gt_col = gt_col.clone()
    dev = Et.device
    cap = (int(pair_cap) if pair_cap is not None else default_pair_cap(Nt)) & ~3      # the list is used in groups of four slots
    if cap < 4:
        raise ValueError('pair_cap must be >= 4')
    s_gt64 = torch.empty((Nt + 2,), device=dev, dtype=torch.float64)[:Nt]
    band_t = torch.empty((Nt + 4,), device=dev, dtype=torch.float32)          # (+ slack: the GEMM fetches 16-byte groups)
    band_v = torch.empty((((Nv + 3) & ~3) + (Nv + 63) // 64 + 4,), device=dev, dtype=torch.float32)     # per column, then (16-byte aligned) per 64-column block
    count = torch.empty((Nt,), device=dev, dtype=torch.int32)
    pairs = torch.empty((4 + 2 * cap,), device=dev, dtype=torch.int32)
    lib, h = _context(dev)
    if emit:
        _call('rank_prepare', lib.laff_rank_prepare_emit, h, emit, _ptr(Et), _ptr(Ev), _ptr(T.buf), _ptr(V.buf), Nt, Nv, H, d, PREC[T.precision],
              float(T.prescale), _ptr(gt_col), int(col0), _ptr(s_gt64), _ptr(band_t), _ptr(band_v), _ptr(count), _ptr(pairs))
    else:
        _call('rank_prepare', lib.laff_rank_prepare, h, _ptr(Et), _ptr(Ev), _ptr(T.buf), _ptr(V.buf), Nt, Nv, H, d, PREC[T.precision],
              float(T.prescale), _ptr(gt_col), int(col0), _ptr(s_gt64), _ptr(band_t), _ptr(band_v), _ptr(count), _ptr(pairs))
    return RankState(Et, Ev, T, V, H, gt_col, int(col0), s_gt64, band_t, band_v, count, pairs, cap)


def rank_prepare_text(Et, Ev, T, gt_col, col0=0, prescale=None):
    """The TEXT side of laff_rank_prepare alone (laff_rank_prepare_part, sides = 1): exact ground-truth scores of the texts Et against
    the fp32 video rows Ev (gt_col - col0 indexes them; -inf outside) and their error bands.  Returns (s_gt64, band_t)."""
    Et, Ev = _emb3(Et, 'Et'), _emb3(Ev, 'Ev')
    Nt, H, d = Et.shape
    if tuple(Ev.shape[1:]) != (H, d) or T.N != Nt or T.K != H * d:
        raise ValueError('embeddings %s / %s do not match the operand (%d x %d)' % (tuple(Et.shape), tuple(Ev.shape), T.N, T.K))
    _dev(gt_col, 'gt_col', torch.int32)
    if gt_col.numel() != Nt or not gt_col.is_contiguous():
        raise ValueError('gt_col must be a contiguous int32 vector of %d' % Nt)
    dev = Et.device
    s_gt64 = torch.empty((Nt + 2,), device=dev, dtype=torch.float64)[:Nt]
    band_t = torch.empty((Nt + 4,), device=dev, dtype=torch.float32)
    lib, h = _context(dev)
    _call('rank_prepare', lib.laff_rank_prepare_part, h, 1, _ptr(Et), _ptr(Ev), _ptr(T.buf), None, Nt, Ev.shape[0], H, d, PREC[T.precision],
          float(T.prescale), _ptr(gt_col), int(col0), _ptr(s_gt64), _ptr(band_t), None, None, None)
    return s_gt64, band_t


def rank_band_video(Ev, V):
    """The VIDEO side of laff_rank_prepare alone (sides = 2): band_v of the rows Ev / their operand V, in the layout the banded GEMM
    reads (per column, then per 64-column block)."""
    Ev = _emb3(Ev, 'Ev')
    Nv, H, d = Ev.shape
    if V.N != Nv or V.K != H * d:
        raise ValueError('embeddings %s do not match the operand (%d x %d)' % (tuple(Ev.shape), V.N, V.K))
    band_v = torch.empty((((Nv + 3) & ~3) + (Nv + 63) // 64 + 4,), device=Ev.device, dtype=torch.float32)
    lib, h = _context(Ev.device)
    _call('rank_prepare', lib.laff_rank_prepare_part, h, 2, None, _ptr(Ev), None, _ptr(V.buf), 0, Nv, H, d, PREC[V.precision], float(V.prescale),
          None, 0, None, None, _ptr(band_v), None, None)
    return band_v


def banded_state(T, V, heads, gt_col, col0, s_gt64, band_t, band_v, pair_cap=None):
    """A RankState for laff_sim_gemm_banded assembled from parts (no fp32 rows: its list is exported, not resolved here)."""
    _dev(gt_col, 'gt_col', torch.int32)
    if gt_col.data_ptr() % 16:
        gt_col = gt_col.clone()
    dev = T.buf.device
    cap = (int(pair_cap) if pair_cap is not None else default_pair_cap(T.N)) & ~3
    count = torch.zeros((T.N,), device=dev, dtype=torch.int32)
    pairs = torch.empty((4 + 2 * cap,), device=dev, dtype=torch.int32)
    pairs[:4].zero_()
    return RankState(None, None, T, V, heads, gt_col, int(col0), s_gt64, band_t, band_v, count, pairs, cap)


def rank_export_pairs(st, S, bounds, col0, cap):
    """The listed pairs of a banded GEMM, bucketed by the owner of the text row (laff_rank_export_pairs).  bounds: int32 device
    tensor (world + 1).  Returns (out (world, cap, 2) int32 -- unused slots -1 --, fill (world + 1) int32)."""
    world = bounds.numel() - 1
    cap = (int(cap) + 3) & ~3
    dev = st.pairs.device
    out = torch.empty((world, cap, 2), device=dev, dtype=torch.int32)
    fill = torch.empty((world + 1,), device=dev, dtype=torch.int32)
    lds = 0
    if S is not None:
        S, lds = _rows(S, 'S')
    lib, h = _context(dev)
    _call('rank_export', lib.laff_rank_export_pairs, h, _ptr(st.s_gt64), _ptr(st.count), _ptr(S), lds, st.V.N, _ptr(st.pairs), st.pair_cap,
          _ptr(bounds), world, int(col0), _ptr(out), cap, _ptr(fill))
    return out, fill


def rank_resolve_list(Et, Ev, s_gt64, count, lst):
    """laff_rank_resolve on a plain list: lst int32 (4 + 2 n), header {0, 0, n, 4} then n slots {row, col} with unused slots -1 and
    the valid pairs first in every group of four (what concatenated laff_rank_export_pairs buckets are).  count += wins."""
    Et, Ev = _emb3(Et, 'Et'), _emb3(Ev, 'Ev')
    Nt, H, d = Et.shape
    n = (lst.numel() - 4) // 2
    lib, h = _context(Et.device)
    _call('rank_resolve', lib.laff_rank_resolve, h, _ptr(Et), _ptr(Ev), Nt, Ev.shape[0], H, d, _ptr(s_gt64), _ptr(count), None, 0, _ptr(lst), n)
    return count


def sim_gemm_banded(st, want_scores=True, out=None):
    """Second launch: the similarity GEMM with the banded count (state.count, state.pairs are filled).  Returns S or None."""
    T, V = st.T, st.V
    dev = T.buf.device
    S, lds = None, V.N
    if want_scores:
        S = out if out is not None else alloc_scores(T.N, V.N, dev)
        S, lds = _rows(S, 'out')
        if tuple(S.shape) != (T.N, V.N):
            raise ValueError('out must be (%d, %d)' % (T.N, V.N))
    scale = 1.0 / (st.heads * T.prescale * V.prescale)
    lib, h = _context(dev)
    _call('sim_gemm', lib.laff_sim_gemm_banded, h, _ptr(T.buf), _ptr(V.buf), T.N, V.N, T.K, scale, PREC[T.precision], _ptr(S), lds,
          _ptr(st.gt_col), st.col0, _ptr(st.s_gt64), _ptr(st.band_t), _ptr(st.band_v), _ptr(st.count), _ptr(st.pairs), st.pair_cap)
    return S


def rank_resolve(st, S=None):
    """Third launch: exact re-score of the listed pairs; state.count (+ S) are final afterwards."""
    lds = 0
    if S is not None:
        S, lds = _rows(S, 'S')
    Nt, H, d = st.Et.shape
    lib, h = _context(st.Et.device)
    _call('rank_resolve', lib.laff_rank_resolve, h, _ptr(st.Et), _ptr(st.Ev), Nt, st.Ev.shape[0], H, d, _ptr(st.s_gt64), _ptr(st.count),
          _ptr(S), lds, _ptr(st.pairs), st.pair_cap)
    return st.count


def rank_resolve_metrics(st, S=None, out_pinned=None, base=1, ranks_out=None):
    """laff_rank_resolve + the rank metrics in ONE launch (laff_rank_resolve_metrics): the resolve workgroup that finishes last turns
    the final counts into ranks (ranks_out <- count + base) and the seven metrics.
    out_pinned None: synchronises, returns the 7-tuple (RuntimeError if a rank < 1 was flagged: overflowed pair list).
    out_pinned (pinned float64 tensor of >= 8): no sync, capturable; the device writes the 8 doubles into it directly; returns None."""
    lds = 0
    if S is not None:
        S, lds = _rows(S, 'S')
    Nt, H, d = st.Et.shape
    lib, h = _context(st.Et.device)
    if ('metrics', h.value) not in _ctx:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('call ops.ctx_prepare_metrics(device) before capturing a graph (it allocates scratch)')
        ctx_prepare_metrics(st.Et.device)
    ro = _ranks_out(st.count, ranks_out)
    if out_pinned is None:
        out = (C.c_double * 8)()
        _call('rank_resolve', lib.laff_rank_resolve_metrics, h, _ptr(st.Et), _ptr(st.Ev), Nt, st.Ev.shape[0], H, d, _ptr(st.s_gt64),
              _ptr(st.count), _ptr(S), lds, _ptr(st.pairs), st.pair_cap, int(base), _ptr(ro), out, 1)
        return tuple(out)[:7]
    if out_pinned.dtype != torch.float64 or out_pinned.numel() < 8 or not out_pinned.is_pinned():
        raise ValueError('out_pinned must be a pinned float64 tensor of >= 8 elements')
    _call('rank_resolve', lib.laff_rank_resolve_metrics, h, _ptr(st.Et), _ptr(st.Ev), Nt, st.Ev.shape[0], H, d, _ptr(st.s_gt64),
          _ptr(st.count), _ptr(S), lds, _ptr(st.pairs), st.pair_cap, int(base), _ptr(ro), C.c_void_p(out_pinned.data_ptr()), 0)
    return None


def exact_ranks(Et, Ev, T, V, gt_col, want_scores=True, col0=0, pair_cap=None):
    """prepare -> banded GEMM -> resolve on one device.  Returns (S or None, count int32 (Nt,), RankState); ranks = count + 1.
    An overflowing pair list (state.listed_pairs()[1]) poisons count[0]: rank_metrics(base=1) raises on it."""
    st = rank_prepare(Et, Ev, T, V, gt_col, col0, pair_cap)
    S = sim_gemm_banded(st, want_scores)
    rank_resolve(st, S)
    return S, st.count, st


def gather_gt(S, gt_col, col0=0):
    S, lds = _rows(S, 'S')
    _dev(gt_col, 'gt_col', torch.int32)
    out = torch.empty((S.shape[0],), device=S.device, dtype=torch.float32)
    lib, h = _context(S.device)
    _call('gather_gt', lib.laff_gather_gt, h, _ptr(S), S.shape[0], S.shape[1], lds, _ptr(gt_col), col0, _ptr(out))
    return out


def rank_count(S, gt_col, s_gt, col0=0, count=None):
    S, lds = _rows(S, 'S')
    _dev(gt_col, 'gt_col', torch.int32)
    _dev(s_gt, 's_gt')
    acc = count is not None
    if count is None:
        count = torch.empty((S.shape[0],), device=S.device, dtype=torch.int32)
    _dev(count, 'count', torch.int32)
    lib, h = _context(S.device)
    _call('rank_count', lib.laff_rank_count, h, _ptr(S), S.shape[0], S.shape[1], lds, _ptr(gt_col), col0, _ptr(s_gt), _ptr(count),
                              1 if acc else 0)
    return count


def _topk_kp(K):
    return 64 if K <= 64 else (512 if K <= 512 else (2048 if K <= 2048 else (4096 if K <= 4096 else 8192)))


def topk_max_columns(K):
    """columns of one laff_topk_rows call: the row lives in LDS next to the K' selected (key, index) pairs"""
    return (160 * 1024 - _topk_kp(K) * 8 - 1040) // 4


def _topk_call(S, K):
    S, lds = _rows(S, 'S')
    Nt, Nv = S.shape
    idx = torch.empty((Nt, K), device=S.device, dtype=torch.int32)
    val = torch.empty((Nt, K), device=S.device, dtype=torch.float32)
    lib, h = _context(S.device)
    _call('topk_rows', lib.laff_topk_rows, h, _ptr(S), Nt, Nv, lds, int(K), _ptr(idx), _ptr(val))
    return idx, val


def _topk_blocks(blocks, K, Nv):
    """K best of every row over column blocks handed in one at a time as (first column, (Nt, w) score tensor): each block's K best
    are taken (laff_topk_rows) and the lists merged by the same kernel -- block lists are laid out in ascending (score, index) order
    and blocks in column order, so that position order equals index order among equal scores and the reference's tie rule (larger
    index first) carries over.  A block may be overwritten as soon as the next one is asked for."""
    cap = topk_max_columns(K)
    vals, idxs = [], []
    for c0, blk in blocks:
        k = min(K, blk.shape[1])
        i, v = _topk_call(blk, k)
        vals.append(v.flip(1))
        idxs.append((i + c0).flip(1))
    if len(vals) == 1 and vals[0].shape[1] == K:
        return idxs[0].flip(1).contiguous(), vals[0].flip(1).contiguous()
    cand_v, cand_i = torch.cat(vals, dim=1).contiguous(), torch.cat(idxs, dim=1).contiguous()
    while cand_v.shape[1] > cap:             # very wide collections: merge groups of block lists first
        group = max(2, cap // K) * K
        nv, ni = [], []
        for c0 in range(0, cand_v.shape[1], group):
            pv, pi = cand_v[:, c0:c0 + group], cand_i[:, c0:c0 + group]
            k = min(K, pv.shape[1])
            p, v = _topk_call(pv, k)
            nv.append(v.flip(1))
            ni.append(torch.gather(pi, 1, p.long()).flip(1))
        cand_v, cand_i = torch.cat(nv, dim=1).contiguous(), torch.cat(ni, dim=1).contiguous()
    pos, val = _topk_call(cand_v, K)
    return torch.gather(cand_i, 1, pos.long()).contiguous(), val


def topk_rows(S, K):
    """Per row: indices (int32) and scores (fp32) of the K best columns, score descending, ties by larger index first (what the
    reference's `np.argsort(...)[::-1][:K]` yields with a stable sort, predictor.py:53-65).  Any number of columns: a collection that
    does not fit the kernel's LDS-resident row (~36k columns at K <= 2048) is split into column blocks (_topk_blocks).  K <= 8192."""
    S, lds = _rows(S, 'S')
    Nt, Nv = S.shape
    K = int(K)
    if K < 1 or K > Nv or K > 8192:
        raise ValueError('topk_rows: need 1 <= K <= min(Nv, 8192), got K=%d for %d columns' % (K, Nv))
    cap = topk_max_columns(K)
    if Nv <= cap:
        return _topk_call(S, K)
    return _topk_blocks(((c0, S[:, c0:min(Nv, c0 + cap)]) for c0 in range(0, Nv, cap)), K, Nv)


def topk_from_operands(T, V, K, heads=1, block_rows=None, scratch_bytes=192 << 20):
    """The K best videos of every text WITHOUT the (Nt, Nv) score matrix (the reference argsorts all of it to keep 500-2000 columns
    per row, predictor.py:53-65; at 100k x 30k that matrix is 12 GB written and read back).  Rows are independent, so the texts are
    taken in blocks: one block of the text operand is scored against all videos into ONE reusable (block_rows, Nv) buffer
    (laff_sim_gemm) and reduced to its lists right away (laff_topk_rows; collections wider than its LDS row are split by columns and
    merged, see topk_rows).  The default buffer (192 MB) stays inside the 256 MB Infinity Cache: the top-K kernel reads the block
    the GEMM just wrote without going to HBM.  (A running top-K inside the GEMM epilogue is not possible at these K: 64 rows x 2000
    entries x 8 bytes per wavefront against 160 KB of LDS per CU.)  Scores and order are exactly those of
    topk_rows(sim_gemm(T, V), K): an entry of the GEMM does not depend on the block it is computed in.
    Returns (idx int32 (Nt, K), val fp32 (Nt, K))."""
    if T.precision != V.precision or T.K != V.K:
        raise ValueError('operands differ in precision or K')
    Nt, Nv, K = T.N, V.N, int(K)
    if K < 1 or K > Nv or K > 8192:
        raise ValueError('topk_from_operands: need 1 <= K <= min(Nv, 8192), got K=%d for %d columns' % (K, Nv))
    dev = T.buf.device
    pitch = (Nv + 31) & ~31
    if block_rows is None:
        block_rows = max(256, (scratch_bytes // (4 * pitch)) & ~255)
    block_rows = int(min(max(int(block_rows), 1), Nt))
    buf = torch.empty((block_rows, pitch), device=dev, dtype=torch.float32)
    idx = torch.empty((Nt, K), device=dev, dtype=torch.int32)
    val = torch.empty((Nt, K), device=dev, dtype=torch.float32)
    for r0 in range(0, Nt, block_rows):
        r1 = min(Nt, r0 + block_rows)
        S = sim_gemm(T.rows(r0, r1), V, heads, out=buf[:r1 - r0, :Nv])
        i, v = topk_rows(S, K)
        idx[r0:r1], val[r0:r1] = i, v
    return idx, val


def v2t_count(S, grp_off, grp_idx, max_group):
    S, lds = _rows(S, 'S')
    _dev(grp_off, 'grp_off', torch.int32)
    _dev(grp_idx, 'grp_idx', torch.int32)
    count = torch.zeros((S.shape[0],), device=S.device, dtype=torch.int32)
    lib, h = _context(S.device)
    _call('v2t_count', lib.laff_v2t_count, h, _ptr(S), S.shape[0], S.shape[1], lds, _ptr(grp_off), _ptr(grp_idx), int(max_group),
                             _ptr(count))
    return count


def v2t_count_exact(S, st, grp_off, grp_idx, max_group, list_cap=None):
    """laff_v2t_count_exact on the score matrix and RankState of exact_ranks (same operands, col0 == 0): int32 (Nt,) counts of texts
    that beat caption t in the column of its video, decided with the exact fp64 scores.  Synchronises (reads the overflow flag);
    a list that was too small is re-sized once."""
    S, lds = _rows(S, 'S')
    _dev(grp_off, 'grp_off', torch.int32)
    _dev(grp_idx, 'grp_idx', torch.int32)
    Nt, H, d = st.Et.shape
    if tuple(S.shape) != (Nt, st.Ev.shape[0]) or st.col0 != 0:
        raise ValueError('S %s does not belong to this RankState (%d x %d, col0 %d)' % (tuple(S.shape), Nt, st.Ev.shape[0], st.col0))
    count = torch.empty((Nt,), device=S.device, dtype=torch.int32)
    cap = int(list_cap) if list_cap is not None else max(1 << 16, 8 * Nt)
    lib, h = _context(S.device)
    for attempt in range(2):
        lst = torch.empty((4 + 3 * cap,), device=S.device, dtype=torch.int32)
        _call('v2t_count_exact', lib.laff_v2t_count_exact, h, _ptr(S), Nt, S.shape[1], lds, _ptr(grp_off), _ptr(grp_idx), int(max_group),
              _ptr(st.Et), _ptr(st.Ev), H, d, _ptr(st.s_gt64), _ptr(st.band_t), _ptr(st.band_v), _ptr(count), _ptr(lst), cap)
        wanted, overflow = [int(x) & 0xffffffff for x in lst[:2].cpu().tolist()]
        if not overflow:
            return count
        cap = wanted + 1024
    raise RuntimeError('laff_v2t_count_exact: the list of in-band pairs overflowed twice (%d wanted)' % wanted)


def _ranks_out(r, ranks_out):
    if ranks_out is not None:
        _dev(ranks_out, 'ranks_out', torch.int32)
        if ranks_out.numel() != r.numel() or not ranks_out.is_contiguous():
            raise ValueError('ranks_out must be a contiguous int32 vector of %d' % r.numel())
    return ranks_out


def rank_metrics(rank1, base=0, ranks_out=None):
    """(r1, r5, r10, medr, meanr, mir, mAP) from int32 device values r with rank = r + base (1-based ranks: base 0; counts of
    better-scoring videos: base 1); ranks_out optionally receives the ranks.  Synchronises the stream."""
    _dev(rank1, 'rank1', torch.int32)
    out = (C.c_double * 7)()
    lib, h = _context(rank1.device)
    _call('rank_metrics', lib.laff_rank_metrics, h, _ptr(rank1.contiguous()), rank1.numel(), int(base), _ptr(_ranks_out(rank1, ranks_out)), out)
    return tuple(out)


def rank_metrics_async(rank1, out_pinned, base=0, ranks_out=None):
    """Launch the metrics reduction and the 64-byte D2H copy without synchronising (HIP-graph capturable).
    out_pinned: pinned CPU float64 tensor of 8; after a stream sync [:7] are the metrics, [7] != 0 flags a rank < 1."""
    _dev(rank1, 'rank1', torch.int32)
    if out_pinned.dtype != torch.float64 or out_pinned.numel() < 8 or not out_pinned.is_pinned():
        raise ValueError('out_pinned must be a pinned float64 tensor of >= 8 elements')
    lib, h = _context(rank1.device)
    if ('metrics', h.value) not in _ctx:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('call ops.ctx_prepare_metrics(device) before capturing a graph (it allocates scratch)')
        ctx_prepare_metrics(rank1.device)
    _call('rank_metrics', lib.laff_rank_metrics_async, h, _ptr(rank1.contiguous()), rank1.numel(), int(base), _ptr(_ranks_out(rank1, ranks_out)),
          C.c_void_p(out_pinned.data_ptr()))


def ctx_prepare_metrics(device):
    """Allocate the ctx's metrics scratch outside any graph capture (hipMalloc is not capturable)."""
    lib, h = _context(device)
    key = ('metrics', h.value)
    if key not in _ctx:
        r = torch.ones(1, dtype=torch.int32, device=device)
        out = (C.c_double * 7)()
        check(lib.laff_rank_metrics(h, _ptr(r), 1, 0, None, out))
        _ctx[key] = True
