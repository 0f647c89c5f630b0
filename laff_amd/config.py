"""Config object for the hot path: the attribute names the towers read off `config`
(/root/reference/configs/base_config.py, configs/laff.py; list in SURVEY.md section 5).

The reference's config *system* (adjust_parm string decoding, trainer.prepare_config) is out of scope; a checkpoint's
own pickled config object works unchanged as long as it carries these attributes.
"""
import copy


class T2V:
    """Stand-in for the reference's txt2vec objects: only `.ndims` is read on the path."""

    def __init__(self, ndims):
        self.ndims = ndims


class config(object):
    model_name = 'LAFF'
    text_encoding = {
        'bow_encoding': {'name': 'bow_nsw'},
        'w2v_encoding': {'name': 'w2v_nsw'},
        'rnn_encoding': {'name': 'nogru_mean'},
        'bert_encoding': {'name': 'noBert', 'dir_name': 'bert-base-uncased'},
        'CLIP_encoding': {'name': 'ViT-B/32', 'dir_name': 'clip_finetune_8frame_uniform_1103'},
        'NetVLAD_encoding': {'name': 'noNetVLAD'},
    }
    rnn_size = 1024
    bert_size = 768
    bert_transform_batch_norm = True
    bert_transform_dropout = 0
    bert_transform_activation = 'tanh'
    clip_opt = {'size': 512, 'transform_batch_norm': True, 'transform_dropout': 0.0, 'transform_activation': 'tanh',
                'frozen': True, 'vocab_size': 49408}
    NetVLAD_opt = {'num_clusters': 32, 'alpha': 100, 'normalize_pooling': False}
    vis_fc_layers = ['0', 4096]
    txt_fc_layers = [0, 4096]
    batch_norm = False
    dropout = 0.2
    activation = 'tanh'
    measure = 'cosine'
    float16 = False
    attention_l2norm = False
    vis_no_transform = []
    txt_no_transform = []
    txt_attention = 'Multi_head_MyApply_Attention'
    vis_attention = 'Multi_head_MyApply_Attention'
    txt_expert_embedding = {'expert': False, 'l2norm': False}
    vis_expert_embedding = {'expert': False, 'l2norm': False}
    vis_feat_add_concat = False
    multi_head_attention = {'dropout': 0.0, 'heads': 8, 'embed_dim_qkv': 512}
    attention_param_each_head = {'with_ave': False, 'mul': False, 'split_head': True}
    txt_attention_global_decay_rate = 0.8
    vis_attention_global_decay_rate = 0.8
    vid_feats = []
    max_frame = 200
    frame_feat_input = False
    frame_feat_with_video_feat = False
    vid_frame_feats = []
    vis_frame_attention = 'attention_noAveNoAverageMul'
    vis_frame_addFC = False
    t2v_bow = T2V(0)
    t2v_w2v = T2V(0)


def make_config(vid_dims, txt_dims, D=4096, heads=8, model_name='LAFF', vis_no_transform=(), txt_no_transform=(),
                frame_feats=None, **overrides):
    """vid_dims: {feature name: dim} in fusion order; txt_dims: subset of {'rnn','bert','bow','w2v','CLIP'} -> dim."""
    c = config()
    c.model_name = model_name
    c.text_encoding = copy.deepcopy(config.text_encoding)
    te = c.text_encoding
    te['rnn_encoding']['name'] = 'gru_mean' if 'rnn' in txt_dims else 'nogru_mean'
    te['bert_encoding']['name'] = 'bert-base-uncased' if 'bert' in txt_dims else 'noBert'
    te['bow_encoding']['name'] = 'bow_nsw' if 'bow' in txt_dims else 'nobow_nsw'
    te['w2v_encoding']['name'] = 'w2v_nsw' if 'w2v' in txt_dims else 'now2v_nsw'
    te['CLIP_encoding']['name'] = 'ViT-B/32' if 'CLIP' in txt_dims else 'noCLIP'
    c.rnn_size = txt_dims.get('rnn', 0)
    c.bert_size = txt_dims.get('bert', 768)
    c.t2v_bow = T2V(txt_dims.get('bow', 0))
    c.t2v_w2v = T2V(txt_dims.get('w2v', 0))
    c.clip_opt = dict(config.clip_opt, size=txt_dims.get('CLIP', 512))
    fc0 = dict(vid_dims)
    c.vid_feats = list(vid_dims.keys())
    if frame_feats:
        c.frame_feat_input = True
        c.vid_frame_feats = list(frame_feats.keys())
        fc0.update(frame_feats)
    c.vis_fc_layers = [fc0, D]
    c.txt_fc_layers = [0, D]
    c.multi_head_attention = {'dropout': 0.0, 'heads': heads, 'embed_dim_qkv': D // heads}
    c.attention_param_each_head = dict(config.attention_param_each_head)
    c.vis_no_transform = list(vis_no_transform)
    c.txt_no_transform = list(txt_no_transform)
    for k, v in overrides.items():
        if k in ('with_ave', 'mul', 'split_head'):
            c.attention_param_each_head[k] = v
        else:
            setattr(c, k, v)
    return c
