"""Shared helpers for the parity tests: build laff_amd models that mirror the golden fixtures."""
import numpy as np
import torch


def load_sd(model, sd_arrays, device=None):
    """load a fixture state_dict (numpy arrays) with strict=False like predictor.py:167; returns missing/unexpected."""
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_arrays.items()}
    res = model.load_state_dict(sd, strict=False)
    return res


def maxdiff(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


ENC_KEY = {'rnn_encoder': 'rnn_encoding', 'bert_encoder': 'bert_encoding', 'bow_encoder': 'bow_encoding',
           'w2v_encoder': 'w2v_encoding', 'CLIP_encoder': 'CLIP_encoding', 'NetVLAD_encoder': 'NetVLAD_encoding'}


def oracle_towers(model, vis_np, txt_np, rows_t=None):
    """Embeddings of both 'LAFF' towers from the oracle, reading the weights off a laff_amd model's state_dict.
    vis_np / txt_np: dense numpy features (a CSR bow must be densified by the caller).  rows_t: optional text row subset."""
    from oracle import laff_oracle as O
    opt = model.vis_net.opt if hasattr(model.vis_net, 'opt') else model.txt_net.opt
    H = opt.multi_head_attention['heads']
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    vspecs = [O.feature_spec(sd, 'vis_net.VisMutiTransformNet.%s.' % n, vis_np[n], 'tanh', H, n in opt.vis_no_transform)
              for n in opt.vid_feats]
    tspecs = []
    for e in model.txt_net.encoder_name_list:
        x = txt_np[ENC_KEY[e]]
        tspecs.append(O.feature_spec(sd, 'txt_net.transform_layer.%s_transform.' % e, x if rows_t is None else x[rows_t],
                                     'tanh', H, e in opt.txt_no_transform))
    ve = O.fuse_tower(vspecs, O.attention_from_sd(sd, 'vis_net.attention_layer.', H, False, False), H)
    te = O.fuse_tower(tspecs, O.attention_from_sd(sd, 'txt_net.attention_layer.', H, False, False), H)
    return ve, te


def oracle_towers_framelaff(model, frames_np, txt_np, rows_t=None):
    """Embeddings of the synthetic 'FrameLAFF' towers (laff_amd.synth.build_model(frames=F): every video feature is a frame
    tensor, Attention_1(with_ave=False, mul=False) over the zero-padded frames -> (B, 512) -> FC -> tanh -> BN -> LAFF) from the
    oracle.  frames_np: {name: (B, Fmax, 512), 'mask_tensor': ...}."""
    from oracle import laff_oracle as O
    opt = model.vis_net.opt
    H = opt.multi_head_attention['heads']
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    vspecs = []
    for n in opt.vid_frame_feats:
        pre = 'vis_net.frame_attention.%s.0.' % n
        vec = O.frame_attention(frames_np[n], sd[pre + 'embedding_common.0.weight'].reshape(-1), sd[pre + 'embedding_common.0.bias'].reshape(()),
                                False, False, sd[pre + 'global_emb_weight_net.weight'].reshape(()))
        vspecs.append(O.feature_spec(sd, 'vis_net.%s.' % n, vec, 'tanh', H, n in opt.vis_no_transform))
    tspecs = []
    for e in model.txt_net.encoder_name_list:
        x = txt_np[ENC_KEY[e]]
        tspecs.append(O.feature_spec(sd, 'txt_net.transform_layer.%s_transform.' % e, x if rows_t is None else x[rows_t],
                                     'tanh', H, e in opt.txt_no_transform))
    ve = O.fuse_tower(vspecs, O.attention_from_sd(sd, 'vis_net.vis_attention_layer.', H, False, False), H)
    te = O.fuse_tower(tspecs, O.attention_from_sd(sd, 'txt_net.attention_layer.', H, False, False), H)
    return ve, te
