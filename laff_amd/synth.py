"""Deterministic synthetic workloads of BASELINE.json: pre-extracted feature matrices with a
planted 64-d latent so that retrieval quality is far above chance, `gt(t) = t mod Nv`, random-init weights of the
reference architecture (Xavier FC, randomised BN statistics)."""
import numpy as np
import torch
import torch.nn as nn

from .config import make_config
from .model import get_model

#: name -> (Nt, Nv, heads, d, frames per video or 0)   -- BASELINE.json `configs`
WORKLOADS = {
    'c2_10kx3k': (10000, 3000, 1, 512, 0),
    'c3_framelaff_10kx3k': (10000, 3000, 1, 512, 32),
    'c4_40kx10k': (40000, 10000, 1, 512, 0),
    'c5_ml_100kx30k': (100000, 30000, 8, 512, 0),
    'c1_test3k': (59800, 2990, 8, 512, 0),       # MSR-VTT test3k shapes: 2,990 videos x 20 captions, see SPECS
    'tiny_c1': (600, 30, 8, 64, 0),
    'tiny': (512, 192, 1, 512, 0),
    'tiny_ml': (384, 128, 8, 512, 0),
    'tiny_frame': (256, 96, 1, 512, 8),
}
#: workloads whose features are not the uniform 4+4 x feat_dim set.  configs/laff.py of the reference: D = 8 x 512, the
#: fine-tuned CLIP feature of both sides skips the FC (no_transform: tiled over heads + BatchNorm), bow is a sparse count
#: vector over the caption vocabulary (gather-sum FC), gt(t) = t // 20 (every video has 20 consecutive captions).
SPECS = {
    'c1_test3k': dict(vid={'clip_ft': 512, 'x3d': 2048}, vis_no_transform=['clip_ft'], txt={'bow': 7811, 'CLIP': 512},
                      txt_no_transform=['CLIP_encoder'], bow_nnz=(4, 24), caps_per_video=20),
    'tiny_c1': dict(vid={'clip_ft': 64, 'x3d': 96}, vis_no_transform=['clip_ft'], txt={'bow': 333, 'CLIP': 64},
                    txt_no_transform=['CLIP_encoder'], bow_nnz=(1, 9), caps_per_video=20),
}
VID_FEATS = ('clip_ft', 'x3d', 'ircsn', 'tf')
TXT_FEATS = ('bow', 'w2v', 'rnn', 'CLIP')       # encoder order in the tower: rnn, bow, w2v, CLIP
TXT_KEY = {'bow': 'bow_encoding', 'w2v': 'w2v_encoding', 'rnn': 'rnn_encoding', 'CLIP': 'CLIP_encoding'}
LATENT = 64


def _randomise_bn(model, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm1d):
            n = m.num_features
            m.weight.data.copy_(torch.empty(n).uniform_(0.5, 1.5, generator=g))
            m.bias.data.copy_(torch.empty(n).normal_(0, 0.1, generator=g))
            m.running_mean.data.copy_(torch.empty(n).normal_(0, 0.1, generator=g))
            m.running_var.data.copy_(torch.empty(n).uniform_(0.5, 1.5, generator=g))


def build_spec_model(spec, heads, d, device, seed=1234):
    """'LAFF' with the feature set of SPECS[...] (mixed dims, no-transform CLIP features, sparse bow)."""
    cfg = make_config(spec['vid'], spec['txt'], heads * d, heads, 'LAFF', vis_no_transform=spec['vis_no_transform'],
                      txt_no_transform=spec['txt_no_transform'], batch_norm=True)
    torch.manual_seed(seed)
    model = get_model('LAFF', device, cfg).eval()
    _randomise_bn(model, seed + 1)
    # the two CLIP features live in one space (fine-tuned jointly): share their BatchNorm and the attention parameters so
    # that the planted latent survives the random-init towers; the x3d / bow branches stay independent (they act as noise)
    model.txt_net.transform_layer.CLIP_encoder_transform.load_state_dict(
        getattr(model.vis_net.VisMutiTransformNet, spec['vis_no_transform'][0]).state_dict())
    model.txt_net.attention_layer.load_state_dict(model.vis_net.attention_layer.state_dict())
    return model


def make_spec_features(spec, Nt, Nv, device, seed=1234, noise=None, sparse_bow=True):
    """(vis{name: (Nv, D_k)}, txt{key: (Nt, D_k)}, gt).  bow is a torch CSR matrix (int32 indices) unless sparse_bow=False."""
    import os
    if noise is None:
        noise = float(os.environ.get('LAFF_SYNTH_NOISE', '1.6'))
    g = torch.Generator(device=device).manual_seed(seed)

    def randn(*shape):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32)

    zv = randn(Nv, LATENT)
    gt = torch.clamp(torch.arange(Nt, device=device, dtype=torch.int64) // spec['caps_per_video'], max=Nv - 1)
    zt = zv[gt]
    vis, txt = {}, {}
    clip_dim = spec['txt']['CLIP']
    P_clip = randn(LATENT, clip_dim) / LATENT ** 0.5
    for n, dim in spec['vid'].items():
        P = P_clip if n in spec['vis_no_transform'] else randn(LATENT, dim) / LATENT ** 0.5
        vis[n] = zv @ P + noise * randn(Nv, dim)
    txt['CLIP_encoding'] = zt @ P_clip + noise * randn(Nt, clip_dim)
    # bag of words: nnz distinct word ids per caption (Zipf-like popularity), counts 1 or 2
    vocab = spec['txt']['bow']
    lo, hi = spec['bow_nnz']
    nnz = torch.randint(lo, hi + 1, (Nt,), generator=g, device=device)
    crow = torch.zeros(Nt + 1, dtype=torch.int64, device=device)
    crow[1:] = torch.cumsum(nnz, 0)
    total = int(crow[-1])
    u = torch.rand(total, generator=g, device=device)
    col = torch.clamp((vocab ** u - 1.0).to(torch.int64), 0, vocab - 1)          # log-uniform ~ Zipf popularity
    val = 1.0 + (torch.rand(total, generator=g, device=device) < 0.1).to(torch.float32)
    row = torch.repeat_interleave(torch.arange(Nt, device=device), nnz)
    dense = torch.zeros((Nt, vocab), dtype=torch.float32, device=device)
    dense.index_put_((row, col), val, accumulate=True)                            # repeated ids add up, as a count vector does
    if sparse_bow:
        sp = dense.to_sparse_csr()
        txt['bow_encoding'] = torch.sparse_csr_tensor(sp.crow_indices().to(torch.int32), sp.col_indices().to(torch.int32),
                                                      sp.values(), size=sp.shape)
    else:
        txt['bow_encoding'] = dense
    return vis, txt, gt.to(torch.int32), None


def build_model(heads, d, device, frames=0, feat_dim=512, seed=1234, spec=None):
    """'LAFF' (or 'FrameLAFF') with 4 video + 4 text features of `feat_dim`; every feature goes through FC->tanh.
    spec: a SPECS entry for a non-uniform feature set."""
    if spec is not None:
        return build_spec_model(spec, heads, d, device, seed)
    D = heads * d
    vid = {n: feat_dim for n in VID_FEATS}
    txt = {n: feat_dim for n in TXT_FEATS}
    if frames:
        cfg = make_config({}, txt, D, heads, 'FrameLAFF', vis_no_transform=[], frame_feats={n: feat_dim for n in VID_FEATS},
                          frame_feat_with_video_feat=False, vis_frame_attention='attention_noAveNoAverageMul',
                          vis_frame_addFC=False, batch_norm=True)
        name = 'FrameLAFF'
    else:
        cfg = make_config(vid, txt, D, heads, 'LAFF', batch_norm=True)
        name = 'LAFF'
    torch.manual_seed(seed)
    model = get_model(name, device, cfg).eval()
    # Random-init towers put text and video in unrelated spaces (chance-level recall).  Tie the text tower to the
    # video tower (feature k <-> feature k, same FC and attention parameters) so that the planted latent survives
    # and R@K / MedR are informative about the arithmetic; the architecture and the work per pair are unchanged.
    # (FrameLAFF: the video-level vector of feature k is the frame-attention output, a unit-norm weighted mean of frames.)
    vis_mods = [getattr(model.vis_net if frames else model.vis_net.VisMutiTransformNet, n) for n in VID_FEATS]
    txt_mods = [getattr(model.txt_net.transform_layer, e + '_transform') for e in model.txt_net.encoder_name_list]
    for tm, vm in zip(txt_mods, vis_mods):
        tm.load_state_dict(vm.state_dict())
    model.txt_net.attention_layer.load_state_dict((model.vis_net.vis_attention_layer if frames else model.vis_net.attention_layer).state_dict())
    _randomise_bn(model, seed + 1)
    return model


def make_features(Nt, Nv, device, frames=0, feat_dim=512, seed=1234, noise=None, spec=None):
    """Returns (vis_feats{name: (Nv, D_k)} or frame tensors, txt_feats{key: (Nt, D_k)}, gt int32 (Nt,), lens|None)."""
    import os
    if spec is not None:
        return make_spec_features(spec, Nt, Nv, device, seed, noise)
    if noise is None:
        noise = float(os.environ.get('LAFF_SYNTH_NOISE', '1.6'))
    g = torch.Generator(device=device).manual_seed(seed)

    def randn(*shape):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32)

    zv = randn(Nv, LATENT)
    gt = torch.arange(Nt, device=device, dtype=torch.int64) % Nv
    zt = zv[gt]
    vis, txt, lens = {}, {}, None
    if frames:
        lens = torch.randint(max(1, frames // 4), frames + 1, (Nv,), generator=g, device=device, dtype=torch.int32)
        mask = (torch.arange(frames, device=device)[None, :] < lens[:, None]).to(torch.float32)
        frame_P = []
        for n in VID_FEATS:
            P = randn(LATENT, feat_dim) / LATENT ** 0.5
            frame_P.append(P)
            f = (zv @ P)[:, None, :] + noise * 2 * randn(Nv, frames, feat_dim)
            vis[n] = (f * mask[:, :, None]).contiguous()
        vis['mask_tensor'] = mask
    else:
        for n in VID_FEATS:
            P = randn(LATENT, feat_dim) / LATENT ** 0.5
            vis[n] = zv @ P + noise * randn(Nv, feat_dim)
    Ps = []
    if not frames:
        g2 = torch.Generator(device=device).manual_seed(seed)      # replay the video projections P_k
        torch.randn(Nv, LATENT, generator=g2, device=device)
        for n in VID_FEATS:
            Ps.append(torch.randn(LATENT, feat_dim, generator=g2, device=device) / LATENT ** 0.5)
            torch.randn(Nv, feat_dim, generator=g2, device=device)
    # text feature k of the tower (encoder order rnn, bow, w2v, CLIP) shares the projection of video feature k
    for i, n in enumerate(('rnn', 'bow', 'w2v', 'CLIP')):
        if frames:      # the matching video-level vector is unit-norm (frame attention output): same scale on the text side
            x = zt @ frame_P[i] + noise * randn(Nt, feat_dim)
            txt[TXT_KEY[n]] = x / x.norm(dim=1, keepdim=True)
        else:
            txt[TXT_KEY[n]] = zt @ Ps[i] + noise * randn(Nt, feat_dim)
    return vis, txt, gt.to(torch.int32), lens


def slice_rows(x, lo, hi):
    """Rows [lo, hi) of a dense tensor or of a CSR matrix (the rank's shard of a feature)."""
    if x.layout == torch.strided:
        return x[lo:hi].contiguous()
    crow, col, val = x.crow_indices(), x.col_indices(), x.values()
    a, b = int(crow[lo]), int(crow[hi])
    return torch.sparse_csr_tensor((crow[lo:hi + 1] - a).to(crow.dtype), col[a:b].contiguous(), val[a:b].contiguous(),
                                   size=(hi - lo, x.shape[1]))


def to_numpy_dict(d):
    return {k: (v.to_dense() if v.layout != torch.strided else v).detach().cpu().numpy() for k, v in d.items()}
