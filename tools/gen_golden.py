#!/usr/bin/env python3
"""Generate golden vectors for the LAFF hot path by running the REAL reference.

Runs only in the build container: it imports the reference's Python from
/root/reference (read-only) with the third-party modules that are absent here
stubbed out, feeds it seeded synthetic inputs and writes
inputs + expected outputs as small .npz / .json fixtures under tests/golden/.
Nothing of the reference (source, bytecode, pickled classes) is written into
the repo: fixtures hold arrays, ids and scalars only.

    python tools/gen_golden.py            # regenerate every fixture
    python tools/gen_golden.py --only eval,bigfile

The reference symbols exercised (file:line relative to /root/reference):
  model/Attention.py:40-105   Attention_1
  model/Attention.py:473-552  Multi_head_MyApply_Attention
  model/Attention.py:26-37    JustAverage
  model/model.py:211-276      TransformNet
  model/model.py:1787-1881    VisMutiTransformNet(+AddAttnetion)
  model/model.py:1641-1709    MultiScaleTxtEncoderAttention
  model/model.py:2101-2194    VisMutiTransformNetPlusFrameFeat
  model/model.py:1003-1079    W2VVPP.get_txt2vis_matrix / predict
  loss.py:8-34                l2norm / cosine_sim
  evaluation.py:11-109        l2norm / cosine_sim / eval_qry2retro / eval
  predictor.py:232-270        argsort + label-matrix loop (re-enacted here with
                              the same numpy calls, then fed to evaluation.eval)
  bigfile.py:13-240           BigFile
"""
import argparse
import copy
import json
import os
import sys
import types

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def import_reference():
    """Stub recipe of SURVEY.md Appendix B."""
    sys.dont_write_bytecode = True
    os.environ.setdefault('HOME', '/tmp')
    sys.path.insert(0, REF)
    import torch  # noqa: F401
    import transformers  # noqa: F401  (must be first: lazy-import machinery)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

    tv = stub('torchvision')
    tv.datasets = stub('torchvision.datasets', Kinetics400=type('Kinetics400', (object,), {}))
    tv.transforms = stub('torchvision.transforms', **{n: _Dummy for n in [
        'Compose', 'Resize', 'CenterCrop', 'TenCrop', 'Lambda', 'ToTensor', 'Normalize',
        'RandomResizedCrop', 'InterpolationMode']})
    stub('prefetch_generator', BackgroundGenerator=_Dummy)
    stub('ftfy', fix_text=lambda s: s)
    stub('nltk', word_tokenize=lambda s: s.split(), pos_tag=lambda s: s)
    stub('nltk.stem', WordNetLemmatizer=_Dummy)
    stub('nltk.corpus', stopwords=_Dummy(), wordnet=_Dummy())
    stub('torch.utils.tensorboard', SummaryWriter=_Dummy)
    import model.model as mm
    mm.clip.load = lambda *a, **k: (None, None)
    return mm


mm = import_reference()
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import loss as ref_loss  # noqa: E402
import evaluation as ref_eval  # noqa: E402
import bigfile as ref_bigfile  # noqa: E402
from model import Attention as ref_att  # noqa: E402

mm.device = torch.device('cpu')
mm.float16 = False
torch.set_grad_enabled(False)


def rng(seed):
    return np.random.default_rng(seed)


def f32(a):
    return np.asarray(a, dtype=np.float32)


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


def randomize_bn(module, g):
    """BatchNorm running stats and affine parameters away from their identity defaults."""
    for m in module.modules():
        if isinstance(m, nn.BatchNorm1d):
            n = m.num_features
            m.weight.data = torch.from_numpy(f32(g.uniform(0.5, 1.5, n)))
            m.bias.data = torch.from_numpy(f32(g.normal(0, 0.1, n)))
            m.running_mean.data = torch.from_numpy(f32(g.normal(0, 0.1, n)))
            m.running_var.data = torch.from_numpy(f32(g.uniform(0.5, 1.5, n)))


def randomize_linear_bias(module, g, scale=0.05):
    for m in module.modules():
        if isinstance(m, nn.Linear) and m.bias is not None:
            m.bias.data = torch.from_numpy(f32(g.normal(0, scale, m.bias.shape)))


def sd_arrays(module, prefix='sd/'):
    out = {}
    for k, v in module.state_dict().items():
        out[prefix + k] = v.detach().cpu().numpy()
    return out


# ----------------------------------------------------------------------------------------------
# (1) Attention_1
# ----------------------------------------------------------------------------------------------
def gen_attention_1():
    g = rng(101)
    arrays = {}
    cases = []
    x = f32(g.normal(0, 1, (37, 5, 512)))
    arrays['x'] = x
    idx = 0
    for with_ave in (False, True):
        for mul in (False, True):
            for gw in (1.0, 0.6, 0.0):
                if not with_ave and gw != 1.0:
                    continue
                torch.manual_seed(7 + idx)
                att = ref_att.Attention_1(512, with_ave=with_ave, mul=mul).eval()
                att.embedding_common[0].bias.data.fill_(float(g.normal(0, 0.3)))
                att.change_raw_global_emb_weight(gw)
                out = att(torch.from_numpy(x))
                key = 'c%d' % idx
                arrays[key + '/w'] = att.embedding_common[0].weight.detach().numpy().reshape(-1)
                arrays[key + '/b'] = att.embedding_common[0].bias.detach().numpy().reshape(())
                arrays[key + '/out'] = out.numpy()
                arrays[key + '/weights'] = att.weights.detach().numpy()
                cases.append({'key': key, 'with_ave': with_ave, 'mul': mul, 'gw': gw})
                idx += 1
    # L=1 and L=8 edge cases (no ave, no mul)
    for L in (1, 8):
        xl = f32(g.normal(0, 1, (11, L, 512)))
        torch.manual_seed(50 + L)
        att = ref_att.Attention_1(512, with_ave=False, mul=False).eval()
        out = att(torch.from_numpy(xl))
        key = 'L%d' % L
        arrays[key + '/x'] = xl
        arrays[key + '/w'] = att.embedding_common[0].weight.detach().numpy().reshape(-1)
        arrays[key + '/b'] = att.embedding_common[0].bias.detach().numpy().reshape(())
        arrays[key + '/out'] = out.numpy()
        cases.append({'key': key, 'with_ave': False, 'mul': False, 'gw': 1.0, 'own_x': True})
    # JustAverage
    arrays['just_average/out'] = ref_att.JustAverage()(torch.from_numpy(x)).numpy()
    arrays['cases'] = np.array(json.dumps(cases))
    save('attention_1', **arrays)


# ----------------------------------------------------------------------------------------------
# (2) Multi_head_MyApply_Attention
# ----------------------------------------------------------------------------------------------
def gen_multi_head():
    g = rng(202)
    arrays = {}
    cases = []
    idx = 0
    for (D, H, split, l2each, with_ave, mul, gw, N, L) in [
        (4096, 8, True, False, False, False, 1.0, 9, 4),
        (4096, 8, True, True, True, False, 0.6, 9, 4),
        (4096, 8, True, False, True, True, 0.8, 7, 3),
        (512, 8, True, False, False, False, 1.0, 13, 5),   # d = 64
        (512, 4, False, False, False, False, 1.0, 6, 3),   # no split: every head sees all of D
        (2048, 1, True, False, True, False, 0.0, 5, 8),    # single head, d = 2048, L = 8
    ]:
        d = D // H
        torch.manual_seed(300 + idx)
        net = ref_att.Multi_head_MyApply_Attention(D, H, d, with_ave=with_ave, mul=mul, split_head=split,
                                                   l2norm_each_head=l2each).eval()
        for h in range(H):
            net.attention_layer[h].embedding_common[0].bias.data.fill_(float(g.normal(0, 0.3)))
        net.change_raw_global_emb_weight(gw)
        x = f32(g.normal(0, 1, (N, L, D)))
        out = net(torch.from_numpy(x))
        key = 'c%d' % idx
        arrays[key + '/x'] = x
        arrays[key + '/out'] = out.numpy()
        arrays.update(sd_arrays(net, key + '/sd/'))
        cases.append({'key': key, 'D': D, 'H': H, 'split_head': split, 'l2norm_each_head': l2each,
                      'with_ave': with_ave, 'mul': mul, 'gw': gw})
        idx += 1
    arrays['cases'] = np.array(json.dumps(cases))
    save('multi_head', **arrays)


# ----------------------------------------------------------------------------------------------
# (3) TransformNet
# ----------------------------------------------------------------------------------------------
def gen_transform_net():
    g = rng(303)
    arrays = {}
    cases = []
    idx = 0
    for (Dk, D, fc, act, bn, N) in [
        (96, 512, True, 'tanh', False, 33),
        (96, 512, True, 'tanh', True, 33),
        (200, 256, True, None, True, 17),
        (77, 130, True, 'relu', False, 70),     # ragged everything
        (64, 192, True, 'sigmoid', True, 5),
        (512, 512, False, False, True, 21),     # the no-transform branch (model/model.py:1803-1805)
        (1030, 260, True, 'tanh', False, 130),
    ]:
        torch.manual_seed(400 + idx)
        net = mm.TransformNet((Dk, D), None, dropout=0.2, batch_norm=bn, activation=act, fc=fc).eval()
        randomize_bn(net, g)
        randomize_linear_bias(net, g)
        x = f32(g.normal(0, 1, (N, Dk if fc else D)))
        y = net(torch.from_numpy(x))
        key = 'c%d' % idx
        arrays[key + '/x'] = x
        arrays[key + '/y'] = y.numpy()
        arrays.update(sd_arrays(net, key + '/sd/'))
        cases.append({'key': key, 'Dk': Dk, 'D': D, 'fc': fc, 'activation': act if act else None, 'batch_norm': bn})
        idx += 1
    arrays['cases'] = np.array(json.dumps(cases))
    save('transform_net', **arrays)


# ----------------------------------------------------------------------------------------------
# model-level helpers
# ----------------------------------------------------------------------------------------------
class PreExtracted(nn.Module):
    """Stands in for a reference text encoder: returns a pre-extracted feature matrix."""

    def __init__(self, key):
        super().__init__()
        self.key = key

    def forward(self, caption_feat_dict, task3=False):
        return {'text_features': caption_feat_dict[self.key]}


class T2V:
    def __init__(self, n):
        self.ndims = n

    def encoding(self, c):
        raise RuntimeError('text features are pre-extracted in the fixtures')


def laff_cfg(vid_dims, txt_dims, D, H, with_ave, mul, batch_norm, vis_no_transform, clip_no_transform=True,
             gw=None):
    """configs/laff.py + the fields trainer.prepare_config (trainer.py:126-212) would fill in."""
    from configs.laff import config as LaffCfg
    cfg = LaffCfg()
    cfg.text_encoding = copy.deepcopy(cfg.text_encoding)
    cfg.adjust_parm('0_12_0_12_%d_%d_1' % (int(with_ave), int(mul)))
    cfg.attention_param_each_head = {'with_ave': with_ave, 'mul': mul, 'split_head': True}
    cfg.multi_head_attention = {'dropout': 0.0, 'heads': H, 'embed_dim_qkv': D // H}
    cfg.batch_norm = batch_norm
    cfg.vid_feats = list(vid_dims.keys())
    cfg.vis_fc_layers = [dict(vid_dims), D]
    cfg.txt_fc_layers = [0, D]
    cfg.vis_no_transform = list(vis_no_transform)
    cfg.txt_no_transform = ['CLIP_encoder'] if clip_no_transform else []
    cfg.clip_opt = dict(cfg.clip_opt)
    cfg.clip_opt['size'] = txt_dims.get('CLIP', 512)
    te = cfg.text_encoding
    te['rnn_encoding']['name'] = 'nogru_mean'
    te['bow_encoding']['name'] = 'bow_nsw' if 'bow' in txt_dims else 'nobow_nsw'
    te['w2v_encoding']['name'] = 'w2v_nsw' if 'w2v' in txt_dims else 'now2v_nsw'
    te['CLIP_encoding']['name'] = 'ViT-B/32' if 'CLIP' in txt_dims else 'noCLIP'
    cfg.t2v_bow = T2V(txt_dims.get('bow', 0))
    cfg.t2v_w2v = T2V(txt_dims.get('w2v', 0))
    cfg.rnn_size = 0
    return cfg


def plug_text_encoders(model):
    for name in model.txt_net.encoder_name_list:
        key = {'bow_encoder': 'bow_feature', 'w2v_encoder': 'w2v_feature', 'CLIP_encoder': 'CLIP_encoding'}[name]
        setattr(model.txt_net.encoder, name, PreExtracted(key))


def randomize_model(model, g, gw_vis=None, gw_txt=None):
    randomize_bn(model, g)
    randomize_linear_bias(model, g)
    for net, gw in ((model.vis_net, gw_vis), (model.txt_net, gw_txt)):
        for attr in ('attention_layer', 'vis_attention_layer'):
            if hasattr(net, attr) and gw is not None:
                getattr(net, attr).change_raw_global_emb_weight(gw)


def model_sd_arrays(model, prefix='sd/'):
    out = {}
    for k, v in model.state_dict().items():
        if k.startswith('txt_net.encoder.'):
            continue
        out[prefix + k] = v.detach().cpu().numpy()
    return out


# ----------------------------------------------------------------------------------------------
# (4) full 'LAFF' towers at the C1 feature mix
# ----------------------------------------------------------------------------------------------
def gen_laff_towers():
    g = rng(404)
    arrays = {}
    cases = []
    for idx, (with_ave, mul, bn, gwv, gwt) in enumerate([(False, False, False, None, None),
                                                        (True, False, True, 0.6, 0.8)]):
        vid_dims = {'clip_finetune_8frame_uniform_1103': 512, 'X3D_L': 40}
        txt_dims = {'bow': 30, 'CLIP': 512}
        cfg = laff_cfg(vid_dims, txt_dims, 4096, 8, with_ave, mul, bn, ['clip_finetune_8frame_uniform_1103'])
        torch.manual_seed(500 + idx)
        model = mm.get_model('LAFF', torch.device('cpu'), cfg).eval()
        plug_text_encoders(model)
        randomize_model(model, g, gwv, gwt)
        N = 24
        vis = {k: f32(g.normal(0, 1, (N, d))) for k, d in vid_dims.items()}
        txt = {'bow_feature': f32((g.uniform(0, 1, (N, 30)) < 0.1).astype(np.float32)),
               'CLIP_encoding': f32(g.normal(0, 1, (N, 512)))}
        vis_in = {k: torch.from_numpy(v.copy()) for k, v in vis.items()}
        vis_emb = model.vis_net(vis_in)
        cap = {'caption': ['%d' % i for i in range(N)]}
        cap.update({k: torch.from_numpy(v.copy()) for k, v in txt.items()})
        txt_emb = model.txt_net(cap)
        scores = model.get_txt2vis_matrix(txt_emb, vis_emb)
        key = 'c%d' % idx
        for k, v in vis.items():
            arrays[key + '/vis/' + k] = v
        for k, v in txt.items():
            arrays[key + '/txt/' + k] = v
        arrays[key + '/vis_emb'] = vis_emb.numpy()
        arrays[key + '/txt_emb'] = txt_emb.numpy()
        arrays[key + '/scores'] = scores.numpy()
        arrays.update(model_sd_arrays(model, key + '/sd/'))
        cases.append({'key': key, 'vid_dims': vid_dims, 'txt_dims': txt_dims, 'D': 4096, 'H': 8,
                      'with_ave': with_ave, 'mul': mul, 'batch_norm': bn,
                      'vis_no_transform': ['clip_finetune_8frame_uniform_1103'], 'txt_no_transform': ['CLIP_encoder'],
                      'encoder_name_list': list(model.txt_net.encoder_name_list)})
    arrays['cases'] = np.array(json.dumps(cases))
    save('laff_towers', **arrays)


# ----------------------------------------------------------------------------------------------
# (4b) 'LAFF' towers with expert embeddings (model/model.py:1848-1873, :1653-1694): add, and add + l2norm(dim=2)
# ----------------------------------------------------------------------------------------------
def gen_laff_expert():
    g = rng(414)
    arrays = {}
    cases = []
    for idx, (l2, bn, with_ave) in enumerate([(False, False, False), (True, True, False), (True, False, True)]):
        vid_dims = {'clip_finetune_8frame_uniform_1103': 128, 'X3D_L': 40, 'irCSN_152_ig65m_16frms': 24}
        txt_dims = {'bow': 30, 'w2v': 20, 'CLIP': 128}
        cfg = laff_cfg(vid_dims, txt_dims, 512, 4, with_ave, False, bn, ['clip_finetune_8frame_uniform_1103'])
        cfg.vis_expert_embedding = {'expert': True, 'l2norm': l2}
        cfg.txt_expert_embedding = {'expert': True, 'l2norm': l2}
        torch.manual_seed(520 + idx)
        model = mm.get_model('LAFF', torch.device('cpu'), cfg).eval()
        plug_text_encoders(model)
        randomize_model(model, g, 0.7 if with_ave else None, 0.4 if with_ave else None)
        N = 19
        vis = {k: f32(g.normal(0, 1, (N, d))) for k, d in vid_dims.items()}
        txt = {'bow_feature': f32((g.uniform(0, 1, (N, 30)) < 0.1).astype(np.float32)),
               'w2v_feature': f32(g.normal(0, 1, (N, 20))), 'CLIP_encoding': f32(g.normal(0, 1, (N, 128)))}
        vis_emb = model.vis_net({k: torch.from_numpy(v.copy()) for k, v in vis.items()})
        cap = {'caption': ['%d' % i for i in range(N)]}
        cap.update({k: torch.from_numpy(v.copy()) for k, v in txt.items()})
        txt_emb = model.txt_net(cap)
        key = 'c%d' % idx
        for k, v in vis.items():
            arrays[key + '/vis/' + k] = v
        for k, v in txt.items():
            arrays[key + '/txt/' + k] = v
        arrays[key + '/vis_emb'] = vis_emb.detach().numpy()
        arrays[key + '/txt_emb'] = txt_emb.detach().numpy()
        arrays.update(model_sd_arrays(model, key + '/sd/'))
        cases.append({'key': key, 'vid_dims': vid_dims, 'txt_dims': txt_dims, 'D': 512, 'H': 4, 'with_ave': with_ave, 'mul': False,
                      'batch_norm': bn, 'l2norm': l2, 'vis_no_transform': ['clip_finetune_8frame_uniform_1103'],
                      'txt_no_transform': ['CLIP_encoder'], 'encoder_name_list': list(model.txt_net.encoder_name_list)})
    arrays['cases'] = np.array(json.dumps(cases))
    save('laff_expert', **arrays)


# ----------------------------------------------------------------------------------------------
# (5) 'FrameLAFF' tower, ragged frame counts
# ----------------------------------------------------------------------------------------------
def framelaff_cfg(vid_dims, frame_feats, D, H, frame_attention, addFC, batch_norm, with_video_feat=True):
    from configs.FrameLaff_NoFrameFc_StrongCLIP_adjust import config as FCfg
    from configs.base_config import config as Base
    cfg = FCfg()
    cfg.text_encoding = copy.deepcopy(cfg.text_encoding)
    te = cfg.text_encoding
    te['rnn_encoding']['name'] = 'nogru_mean'
    te['bow_encoding']['name'] = 'bow_nsw'
    te['w2v_encoding']['name'] = 'now2v_nsw'
    te['CLIP_encoding']['name'] = 'ViT-B/32'
    cfg.t2v_bow = T2V(20)
    cfg.t2v_w2v = T2V(0)
    cfg.rnn_size = 0
    cfg.vid_feats = list(vid_dims.keys())
    cfg.vid_frame_feats = list(frame_feats)
    cfg.vis_no_transform = list(frame_feats)
    fc0 = dict(vid_dims)
    for f in frame_feats:
        fc0[f] = 512
    cfg.vis_fc_layers = [fc0, D]
    cfg.txt_fc_layers = [0, D]
    cfg.vis_attention = Base.attention_types[12]
    cfg.txt_attention = Base.attention_types[12]
    cfg.vis_frame_attention = frame_attention
    cfg.vis_frame_addFC = addFC
    cfg.batch_norm = batch_norm
    cfg.frame_feat_with_video_feat = with_video_feat
    cfg.attention_param_each_head = {'with_ave': False, 'mul': False, 'split_head': True}
    cfg.multi_head_attention = {'dropout': 0.0, 'heads': H, 'embed_dim_qkv': D // H}
    return cfg


def gen_framelaff():
    g = rng(505)
    arrays = {}
    cases = []
    variants = [
        ('attention_noAveNoAverageMul', False, True, True),    # shipped LAFF-ml: '0_7_1_12_0_12_0'
        ('average_AverageMul_noAve', False, True, True),       # config-file default: mul over frames
        ('attention_noAveNoAverageMul', True, False, True),    # Linear(512,512) in front of the frame attention
        ('attention_noAverageMul_Ave', False, True, False),    # with_ave over frames; frame feats only
    ]
    for idx, (fatt, addFC, bn, with_vid) in enumerate(variants):
        vid_dims = {'X3D_L': 40, 'mean_irCSN': 24}
        frame_feats = ['Frame_clip_ft']
        D, H = (4096, 8) if idx == 0 else (1024, 2)
        cfg = framelaff_cfg(vid_dims, frame_feats, D, H, fatt, addFC, bn, with_vid)
        torch.manual_seed(600 + idx)
        model = mm.get_model('FrameLAFF', torch.device('cpu'), cfg).eval()
        plug_text_encoders(model)
        randomize_model(model, g)
        if fatt == 'attention_noAverageMul_Ave':
            model.vis_net.frame_attention['Frame_clip_ft'][-1].change_raw_global_emb_weight(0.7)
        B, Fmax = 10, 13
        lens = g.integers(3, Fmax + 1, B)
        lens[0] = Fmax
        lens[3] = 1
        frames = np.zeros((B, Fmax, 512), np.float32)
        mask = np.zeros((B, Fmax), np.float32)
        for b in range(B):
            frames[b, :lens[b]] = f32(g.normal(0, 1, (lens[b], 512)))
            mask[b, :lens[b]] = 1
        vis = {k: f32(g.normal(0, 1, (B, d))) for k, d in vid_dims.items()}
        vis_in = {k: torch.from_numpy(v.copy()) for k, v in vis.items()}
        frame_in = {'mask_tensor': torch.from_numpy(mask.copy()), 'Frame_clip_ft': torch.from_numpy(frames.copy())}
        vis_emb = model.vis_net(vis_in, vis_frame_feat_dict_input=frame_in)
        # the per-video aggregated frame feature the tower inserts into vis_input (model/model.py:2171-2173)
        frame_vec = vis_in['Frame_clip_ft'] if with_vid else None
        key = 'c%d' % idx
        for k, v in vis.items():
            arrays[key + '/vis/' + k] = v
        arrays[key + '/frames'] = frames
        arrays[key + '/mask'] = mask
        arrays[key + '/lens'] = lens.astype(np.int32)
        arrays[key + '/vis_emb'] = vis_emb.numpy()
        if frame_vec is not None:
            # after the tower ran, vis_in[feat] holds the tiled (B, 512*H) tensor; first 512 columns = aggregate
            arrays[key + '/frame_vec'] = frame_vec.numpy()[:, :512]
        sd = {k2: v for k2, v in model_sd_arrays(model, key + '/sd/').items() if '/sd/vis_net.' in k2}
        arrays.update(sd)
        cases.append({'key': key, 'vid_dims': vid_dims, 'frame_feats': frame_feats, 'D': D, 'H': H,
                      'vis_frame_attention': fatt, 'vis_frame_addFC': addFC, 'batch_norm': bn,
                      'frame_feat_with_video_feat': with_vid, 'max_frame': Fmax})
    arrays['cases'] = np.array(json.dumps(cases))
    save('framelaff', **arrays)


# ----------------------------------------------------------------------------------------------
# (6) get_txt2vis_matrix 2-D / 3-D, loss.l2norm / cosine_sim
# ----------------------------------------------------------------------------------------------
def gen_txt2vis():
    g = rng(606)
    arrays = {}
    model = mm.W2VVPP_MultiHeadAttention(None)
    t2 = f32(g.normal(0, 1, (41, 512)))
    v2 = f32(g.normal(0, 1, (29, 512)))
    arrays['t2'] = t2
    arrays['v2'] = v2
    arrays['s2'] = model.get_txt2vis_matrix(torch.from_numpy(t2), torch.from_numpy(v2)).numpy()
    t3 = ref_loss.l2norm(torch.from_numpy(f32(g.normal(0, 1, (23, 8, 512)))), dim=2).numpy()
    v3 = ref_loss.l2norm(torch.from_numpy(f32(g.normal(0, 1, (31, 8, 512)))), dim=2).numpy()
    arrays['t3'] = t3
    arrays['v3'] = v3
    arrays['s3'] = model.get_txt2vis_matrix(torch.from_numpy(t3), torch.from_numpy(v3)).numpy()
    # un-normalised 3-D input: cosine_sim re-normalises per head
    t3u = f32(g.normal(0, 2, (7, 4, 64)))
    v3u = f32(g.normal(0, 3, (9, 4, 64)))
    arrays['t3u'] = t3u
    arrays['v3u'] = v3u
    arrays['s3u'] = model.get_txt2vis_matrix(torch.from_numpy(t3u), torch.from_numpy(v3u)).numpy()
    x = f32(g.normal(0, 1, (6, 33)))
    x[2] = 0
    arrays['l2/x'] = x
    arrays['l2/default'] = ref_loss.l2norm(torch.from_numpy(x)).numpy()
    arrays['l2/eps0'] = ref_loss.l2norm(torch.from_numpy(x), eps=0).numpy()
    arrays['l2/np'] = ref_eval.l2norm(x).astype(np.float32)
    arrays['cos/np'] = ref_eval.cosine_sim(t2, v2).astype(np.float32)
    save('txt2vis', **arrays)


# ----------------------------------------------------------------------------------------------
# (7) predict() end to end with fake loaders (uneven last batches, per-batch text re-ordering)
# ----------------------------------------------------------------------------------------------
class _DS:
    def __init__(self, n):
        self.length = n

    def __len__(self):
        return self.length


class FakeVisLoader:
    def __init__(self, feats, ids, bs, frames=None, lens=None):
        self.feats, self.ids, self.batch_size = feats, ids, bs
        self.dataset = _DS(len(ids))
        self.frames, self.lens = frames, lens

    def __len__(self):
        return (len(self.ids) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.ids)
        for s in range(0, n, self.batch_size):
            e = min(n, s + self.batch_size)
            fd = {}
            if self.frames is not None:
                L = self.lens[s:e]
                fmax = int(L.max())
                mask = np.zeros((e - s, fmax), np.float32)
                for i, l in enumerate(L):
                    mask[i, :l] = 1
                fd['mask_tensor'] = torch.from_numpy(mask)
                for k, v in self.frames.items():
                    fd[k] = torch.from_numpy(v[s:e, :fmax].copy())
            yield {'vis_feat_dict': {k: torch.from_numpy(v[s:e].copy()) for k, v in self.feats.items()},
                   'idxs': list(range(s, e)), 'vis_ids': tuple(self.ids[s:e]),
                   'vis_frame_feat_dict': fd, 'vis_origin_frame_tuple': (None,) * (e - s)}


class FakeTxtLoader:
    def __init__(self, feats, ids, bs, perm):
        self.feats, self.ids, self.batch_size, self.perm = feats, ids, bs, perm
        self.dataset = _DS(len(ids))

    def __len__(self):
        return (len(self.ids) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.ids)
        for s in range(0, n, self.batch_size):
            e = min(n, s + self.batch_size)
            order = self.perm[s:e]     # collate_text sorts each batch by token count (data_provider.py:77)
            cap = {'caption': [self.ids[i] for i in order]}
            for k, v in self.feats.items():
                cap[k] = torch.from_numpy(v[order].copy())
            yield cap, [int(i) for i in order], tuple(self.ids[i] for i in order)


def planted(g, Nv, Nt, dims_v, dims_t, latent=16, noise=0.5):
    zv = g.normal(0, 1, (Nv, latent))
    gt = np.arange(Nt) % Nv
    zt = zv[gt] + 0.3 * g.normal(0, 1, (Nt, latent))
    vis = {k: f32(zv @ g.normal(0, 1, (latent, d)) / np.sqrt(latent) + noise * g.normal(0, 1, (Nv, d)))
           for k, d in dims_v.items()}
    txt = {k: f32(zt @ g.normal(0, 1, (latent, d)) / np.sqrt(latent) + noise * g.normal(0, 1, (Nt, d)))
           for k, d in dims_t.items()}
    return vis, txt, gt


def gen_predict():
    g = rng(707)
    arrays = {}
    Nv, Nt, bs = 120, 300, 64
    D, H = 512, 8
    vid_dims = {'clipft': 64, 'x3d': 48, 'ircsn': 40}
    txt_dims = {'bow': 30, 'w2v': 50, 'CLIP': 64}
    cfg = laff_cfg(vid_dims, txt_dims, D, H, False, False, False, ['clipft'])
    torch.manual_seed(700)
    model = mm.get_model('LAFF', torch.device('cpu'), cfg).eval()
    plug_text_encoders(model)
    randomize_model(model, g)
    vis, txt_raw, gt = planted(g, Nv, Nt, vid_dims, txt_dims)
    txt = {'bow_feature': txt_raw['bow'], 'w2v_feature': txt_raw['w2v'], 'CLIP_encoding': txt_raw['CLIP']}
    vis_ids = ['video%d' % i for i in range(Nv)]
    txt_ids = ['video%d#%d' % (gt[i], i // Nv) for i in range(Nt)]
    perm = np.arange(Nt)
    for s in range(0, Nt, bs):
        e = min(Nt, s + bs)
        perm[s:e] = g.permutation(perm[s:e])
    vloader = FakeVisLoader(vis, vis_ids, bs)
    tloader = FakeTxtLoader(txt, txt_ids, bs, perm)
    scores, out_txt_ids, out_vis_ids = model.predict(tloader, vloader, 'cosine', record_emb=True)
    for k, v in vis.items():
        arrays['vis/' + k] = v
    for k, v in txt.items():
        arrays['txt/' + k] = v
    arrays['perm'] = perm.astype(np.int64)
    arrays['scores'] = scores.astype(np.float32)
    arrays['txt_ids_out'] = np.array(json.dumps(list(out_txt_ids)))
    arrays['vis_ids_out'] = np.array(json.dumps(list(out_vis_ids)))
    arrays['txt_ids'] = np.array(json.dumps(txt_ids))
    arrays['vis_ids'] = np.array(json.dumps(vis_ids))
    arrays['video_all_embs'] = model.video_all_embs.numpy()
    arrays.update(model_sd_arrays(model))
    # predictor.py:232-246 re-enacted on the reference's own output, then evaluation.eval
    t2v, v2t = predictor_metrics(scores, list(out_txt_ids), list(out_vis_ids))
    arrays['t2v_metrics'] = np.array(t2v, np.float64)
    arrays['v2t_metrics'] = np.array(v2t, np.float64)
    arrays['cfg'] = np.array(json.dumps({'vid_dims': vid_dims, 'txt_dims': txt_dims, 'D': D, 'H': H, 'bs': bs,
                                         'vis_no_transform': ['clipft'], 'txt_no_transform': ['CLIP_encoder'],
                                         'with_ave': False, 'mul': False, 'batch_norm': False,
                                         'encoder_name_list': list(model.txt_net.encoder_name_list)}))
    save('predict', **arrays)


def predictor_metrics(t2i_matrix, txt_ids, vis_ids):
    """The rank/label arithmetic of predictor.py:232-246 (T2V) and :262-270 (V2T), same numpy calls."""
    inds = np.argsort(t2i_matrix, axis=1)
    label_matrix = np.zeros(inds.shape)
    for index in range(inds.shape[0]):
        ind = inds[index][::-1]
        gt_index = np.where(np.array(vis_ids)[ind] == txt_ids[index].split('#')[0])[0]
        label_matrix[index][gt_index] = 1
    t2v = ref_eval.eval(label_matrix)
    i2t_matrix = t2i_matrix.T
    inds = np.argsort(i2t_matrix, axis=1)
    label_matrix = np.zeros(inds.shape)
    txt_ids2 = [t.split('#')[0] for t in txt_ids]
    for index in range(inds.shape[0]):
        ind = inds[index][::-1]
        label_matrix[index][np.where(np.array(txt_ids2)[ind] == vis_ids[index])[0]] = 1
    v2t = ref_eval.eval(label_matrix)
    return t2v, v2t


# ----------------------------------------------------------------------------------------------
# (8) rank/label -> evaluation.eval, (9) eval_qry2retro
# ----------------------------------------------------------------------------------------------
def gen_eval():
    g = rng(808)
    arrays = {}
    cases = []
    for idx, (Nv, per) in enumerate([(50, 1), (40, 20), (97, 3)]):
        Nt = Nv * per
        gt = np.arange(Nt) % Nv
        S = f32(g.normal(0, 1, (Nt, Nv)))
        S[np.arange(Nt), gt] += f32(g.normal(1.5, 1.0, Nt))      # GT usually, not always, near the top
        vis_ids = ['v%03d' % i for i in range(Nv)]
        txt_ids = ['v%03d#%d' % (gt[i], i // Nv) for i in range(Nt)]
        t2v, v2t = predictor_metrics(S, txt_ids, vis_ids)
        key = 'c%d' % idx
        arrays[key + '/S'] = S
        arrays[key + '/gt'] = gt.astype(np.int32)
        arrays[key + '/t2v'] = np.array(t2v, np.float64)
        arrays[key + '/v2t'] = np.array(v2t, np.float64)
        arrays[key + '/txt_ids'] = np.array(json.dumps(txt_ids))
        arrays[key + '/vis_ids'] = np.array(json.dumps(vis_ids))
        cases.append({'key': key, 'Nv': Nv, 'per': per})
    # a raw label matrix with several ones per row straight into evaluation.eval
    lab = np.zeros((12, 30))
    for r in range(12):
        lab[r, g.choice(30, size=int(g.integers(1, 5)), replace=False)] = 1
    arrays['label/matrix'] = lab
    arrays['label/metrics'] = np.array(ref_eval.eval(lab), np.float64)
    # (9) eval_qry2retro, n_qry = 1
    Sq = f32(g.normal(0, 1, (60, 60)))
    Sq[np.arange(60), np.arange(60)] += 1.0
    arrays['q2r/S'] = Sq
    arrays['q2r/metrics'] = np.array(ref_eval.eval_qry2retro(Sq, 1), np.float64)
    arrays['cases'] = np.array(json.dumps(cases))
    save('eval', **arrays)


# (9b) evaluation.cosine_sim / evaluation.l2norm (numpy API, evaluation.py:11-16, 44-49)
def gen_eval_cosine():
    g = rng(818)
    q = f32(g.normal(0, 1, (37, 96)))
    r = f32(g.normal(0, 1, (53, 96)))
    q[3] = 0.0                      # a zero query: the 1e-10 in the denominator decides what comes out
    q[5] *= 1e-9                    # a tiny one: |x| comparable to the epsilon
    r[7] *= 3e4
    save('eval_cosine', q=q, r=r, sim=np.asarray(ref_eval.cosine_sim(q, r)), l2q=np.asarray(ref_eval.l2norm(q)))


# ----------------------------------------------------------------------------------------------
# (10) BigFile
# ----------------------------------------------------------------------------------------------
def gen_bigfile():
    g = rng(909)
    expect = {}
    for name, sep, n, d in [('bigfile_nl', '\n', 4, 3), ('bigfile_sp', ' ', 6, 5)]:
        ddir = os.path.join(OUT, name)
        os.makedirs(ddir, exist_ok=True)
        ids = ['vid_%c' % (ord('a') + i) for i in range(n)]
        mat = f32(g.normal(0, 1, (n, d)))
        mat.tofile(os.path.join(ddir, 'feature.bin'))
        open(os.path.join(ddir, 'id.txt'), 'w').write(sep.join(ids) + ('\n' if sep == '\n' else ''))
        open(os.path.join(ddir, 'shape.txt'), 'w').write('%d %d' % (n, d))
        bf = ref_bigfile.BigFile(ddir)
        req = [ids[2], 'nope', ids[0], ids[2], ids[-1]]
        names, vecs = bf.read(req)
        by_idx_names, by_idx_vecs = bf.read([3, 1], isname=False)
        err = None
        try:
            bf.read_one('missing')
        except Exception as e:  # noqa: BLE001
            err = type(e).__name__
        expect[name] = {
            'shape': bf.shape(), 'names': list(bf.names), 'ndims': bf.ndims, 'nr_of_images': bf.nr_of_images,
            'request': req, 'read_names': list(names), 'read_vecs': [list(map(float, v)) for v in vecs],
            'read_empty': [list(x) for x in bf.read(['zzz'])],
            'by_index_names': list(by_idx_names), 'by_index_vecs': [list(map(float, v)) for v in by_idx_vecs],
            'read_one': list(map(float, bf.read_one(ids[1]))), 'read_one_missing_error': err,
            'matrix': mat.tolist(),
        }
    json.dump(expect, open(os.path.join(OUT, 'bigfile_expect.json'), 'w'), indent=1)
    print('wrote bigfile fixtures')


# ----------------------------------------------------------------------------------------------
# (11) result writers: predictor.txt2video_write_to_file (predictor.py:53-88)
# ----------------------------------------------------------------------------------------------
def gen_writers():
    import pickle
    import tempfile
    import predictor as ref_predictor
    g = rng(1111)
    arrays = {}
    Nt, Nv = 25, 30
    S = f32(g.normal(0, 0.2, (Nt, Nv)))
    S[3, 7] = S[3, 9]                       # an exact tie inside the kept range
    vis_ids = ['video%d' % i for i in range(Nv)]
    txt_ids = ['video%d#%d' % (i % Nv, i // Nv) for i in range(Nt)]

    class _DS:
        def get_caption_dict_by_id(self, cid):
            return {'caption': 'caption of ' + cid}

    class _TL:
        dataset = _DS()

    inds = np.argsort(S, axis=1, kind='stable')
    for name, thr in (('top10', 10), ('all', 2000)):
        with tempfile.TemporaryDirectory() as d:
            f = os.path.join(d, 'id.sent.score.txt')
            pk = os.path.join(d, 't2v.pkl')
            ref_predictor.txt2video_write_to_file(f, inds, vis_ids, txt_ids, S, pkl_saved_file=pk, txt_loader=_TL(), Threshold=thr)
            arrays[name + '/text'] = np.array(open(f).read())
            dct = pickle.load(open(pk, 'rb'))
            arrays[name + '/pkl'] = np.array(json.dumps({k: {'query': v['query'], 'rank_list': list(v['rank_list']),
                                                             'sim_value': [repr(float(x)) for x in v['sim_value']]}
                                                         for k, v in dct.items()}))
    arrays['S'] = S
    arrays['vis_ids'] = np.array(json.dumps(vis_ids))
    arrays['txt_ids'] = np.array(json.dumps(txt_ids))
    save('writers', **arrays)


# ----------------------------------------------------------------------------------------------
# (12) text-side feature producers: txt2vec.BowVec(NSW) / W2Vec(NSW) + textlib.TextTool.tokenize
# ----------------------------------------------------------------------------------------------
def gen_txt2vec():
    import pickle
    import tempfile
    import textlib as ref_textlib
    import txt2vec as ref_t2v
    g = rng(1212)
    stop = sorted(ref_textlib.ENGLISH_STOP_WORDS)
    content = ['man', 'woman', 'girl', 'young', 'two', 'dog', 'cat', 'car', 'street', 'guitar', 'playing', 'singing', 'dancing',
               'kitchen', 'cooking', 'food', 'talking', 'camera', 'stage', 'crowd', 'game', 'soccer', 'player', 'ball', 'running',
               'water', 'beach', 'baby', 'laughing', 'video', '2', '3d', 'tv', 'news', 'anchor', 'cartoon', 'character', 'minecraft']
    vocab_words = content[:30] + ['the', 'a', 'is', 'on']        # a vocabulary WITH a few stop words (the non-NSW flavour)
    captions = ['A man is playing the guitar on a stage.', 'Two young girls dancing & singing!!', 'a dog, a DOG and a cat',
                'minecraft gameplay video', '', 'the a is on', "someone's cooking food in the kitchen\r\nwhile talking",
                'UNKNOWNWORD zzz', 'man man man woman', 'Soccer-player kicks ball; crowd laughing 3D tv 2', '  water   beach  ',
                'a cartoon character is talking to the camera', 'news anchor', 'baby']
    arrays = {'captions': np.array(json.dumps(captions)), 'stopwords': np.array(json.dumps(stop)),
              'vocab': np.array(json.dumps(vocab_words))}
    for rm in (False, True):
        arrays['tokens_%s' % ('nsw' if rm else 'all')] = np.array(json.dumps(
            [ref_textlib.TextTool.tokenize(c, clean=True, language='en', remove_stopword=rm) for c in captions]))
    arrays['tokens_noclean'] = np.array(json.dumps([ref_textlib.TextTool.tokenize(c, clean=False) for c in captions]))
    with tempfile.TemporaryDirectory() as d:
        voc = ref_textlib.Vocabulary('bow')
        for w in vocab_words:
            voc.add(w)
        for fname, cls, key in (('bow_5.pkl', ref_t2v.BowVec, 'bow'), ('bow_nsw_5.pkl', ref_t2v.BowVecNSW, 'bow_nsw')):
            path = os.path.join(d, fname)
            pickle.dump(voc, open(path, 'wb'))
            t2v = cls(path)
            arrays[key] = np.stack([t2v.encoding(c) for c in captions])          # float64 count vectors
            assert t2v.ndims == len(vocab_words)
        # word2vec table as a BigFile directory
        w2v_words = content[5:38] + ['the', 'on']
        ndims = 20
        table = f32(g.normal(0, 1, (len(w2v_words), ndims)))
        wdir = os.path.join(d, 'vec20')
        os.makedirs(wdir)
        table.tofile(os.path.join(wdir, 'feature.bin'))
        open(os.path.join(wdir, 'id.txt'), 'w').write(' '.join(w2v_words))
        open(os.path.join(wdir, 'shape.txt'), 'w').write('%d %d' % table.shape)
        for cls, key in ((ref_t2v.W2Vec, 'w2v'), (ref_t2v.W2VecNSW, 'w2v_nsw')):
            t2v = cls(wdir)
            arrays[key] = np.stack([t2v.encoding(c) for c in captions])          # float64 means
        arrays['w2v_words'] = np.array(json.dumps(w2v_words))
        arrays['w2v_table'] = table
    save('txt2vec', **arrays)


# ----------------------------------------------------------------------------------------------
# (13) training loss: loss.MarginRankingLoss forward + autograd backward, per head and summed (model.py:2032-2048)
# ----------------------------------------------------------------------------------------------
def gen_margin_loss():
    g = rng(1313)
    arrays = {}
    cases = []
    torch.set_grad_enabled(True)
    try:
        for ci, (B, H, d, margin, maxv, style, direction) in enumerate([
                (24, 1, 32, 0.2, True, 'sum', 't2i'), (24, 1, 32, 0.2, False, 'sum', 'bidir'), (37, 8, 64, 0.2, True, 'sum', 't2i'),
                (37, 8, 64, 0.3, True, 'mean', 'bidir'), (50, 2, 48, 0.1, False, 'mean', 'i2t'), (33, 4, 20, 0.5, True, 'sum', 'i2t'),
                (96, 8, 128, 0.2, True, 'sum', 't2i')]):
            k = 'c%d' % ci
            z = f32(g.normal(0, 1, (B, 16)))                       # shared latent: matched pairs score high, some violations
            P = f32(g.normal(0, 1, (16, H * d)))
            s = torch.tensor(f32(z @ P + 1.5 * g.normal(0, 1, (B, H * d))).reshape(B, H, d), requires_grad=True)
            im = torch.tensor(f32(z @ P + 1.5 * g.normal(0, 1, (B, H * d))).reshape(B, H, d), requires_grad=True)
            crit = ref_loss.MarginRankingLoss(margin=margin, measure='cosine', max_violation=maxv, cost_style=style,
                                              direction=direction)
            total = 0
            for h in range(H):                                   # model/model.py:2037-2039
                total = total + crit(s[:, h, :], im[:, h, :])
            total.backward()
            arrays[k + '/s'] = s.detach().numpy()
            arrays[k + '/im'] = im.detach().numpy()
            arrays[k + '/loss'] = np.float32(total.item())
            arrays[k + '/d_s'] = s.grad.numpy()
            arrays[k + '/d_im'] = im.grad.numpy()
            cases.append(dict(key=k, B=B, H=H, d=d, margin=margin, max_violation=maxv, cost_style=style, direction=direction))
    finally:
        torch.set_grad_enabled(False)
    arrays['cases'] = np.array(json.dumps(cases))
    save('margin_loss', **arrays)


GENERATORS = {
    'margin_loss': gen_margin_loss,
    'txt2vec': gen_txt2vec,
    'attention_1': gen_attention_1, 'multi_head': gen_multi_head, 'transform_net': gen_transform_net,
    'laff_towers': gen_laff_towers, 'laff_expert': gen_laff_expert, 'eval_cosine': gen_eval_cosine, 'framelaff': gen_framelaff, 'txt2vis': gen_txt2vis,
    'predict': gen_predict, 'eval': gen_eval, 'bigfile': gen_bigfile, 'writers': gen_writers,
}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    todo = [s for s in args.only.split(',') if s] or list(GENERATORS)
    for name in todo:
        print('== %s' % name)
        GENERATORS[name]()
