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
    'tiny': (512, 192, 1, 512, 0),
    'tiny_ml': (384, 128, 8, 512, 0),
    'tiny_frame': (256, 96, 1, 512, 8),
}
VID_FEATS = ('clip_ft', 'x3d', 'ircsn', 'tf')
TXT_FEATS = ('bow', 'w2v', 'rnn', 'CLIP')       # encoder order in the tower: rnn, bow, w2v, CLIP
TXT_KEY = {'bow': 'bow_encoding', 'w2v': 'w2v_encoding', 'rnn': 'rnn_encoding', 'CLIP': 'CLIP_encoding'}
LATENT = 64


def build_model(heads, d, device, frames=0, feat_dim=512, seed=1234):
    """'LAFF' (or 'FrameLAFF') with 4 video + 4 text features of `feat_dim`; every feature goes through FC->tanh."""
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
    if not frames:
        # Random-init towers put text and video in unrelated spaces (chance-level recall).  Tie the text tower to the
        # video tower (feature k <-> feature k, same FC and attention parameters) so that the planted latent survives
        # and R@K / MedR are informative about the arithmetic; the architecture and the work per pair are unchanged.
        vis_mods = [getattr(model.vis_net.VisMutiTransformNet, n) for n in VID_FEATS]
        txt_mods = [getattr(model.txt_net.transform_layer, e + '_transform') for e in model.txt_net.encoder_name_list]
        for tm, vm in zip(txt_mods, vis_mods):
            tm.load_state_dict(vm.state_dict())
        model.txt_net.attention_layer.load_state_dict(model.vis_net.attention_layer.state_dict())
    g = torch.Generator(device='cpu').manual_seed(seed + 1)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm1d):
            n = m.num_features
            m.weight.data.copy_(torch.empty(n).uniform_(0.5, 1.5, generator=g))
            m.bias.data.copy_(torch.empty(n).normal_(0, 0.1, generator=g))
            m.running_mean.data.copy_(torch.empty(n).normal_(0, 0.1, generator=g))
            m.running_var.data.copy_(torch.empty(n).uniform_(0.5, 1.5, generator=g))
    return model


def make_features(Nt, Nv, device, frames=0, feat_dim=512, seed=1234, noise=None):
    """Returns (vis_feats{name: (Nv, D_k)} or frame tensors, txt_feats{key: (Nt, D_k)}, gt int32 (Nt,), lens|None)."""
    import os
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
        for n in VID_FEATS:
            P = randn(LATENT, feat_dim) / LATENT ** 0.5
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
        P = Ps[i] if Ps else randn(LATENT, feat_dim) / LATENT ** 0.5
        txt[TXT_KEY[n]] = zt @ P + noise * randn(Nt, feat_dim)
    return vis, txt, gt.to(torch.int32), lens


def to_numpy_dict(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}
