"""Host-side Python of laff_amd (no GPU): evaluation, BigFile, config/model construction, state_dict key parity."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from laff_amd import evaluation
from laff_amd.bigfile import BigFile
from laff_amd.config import make_config
from laff_amd.model import get_model
from laff_amd.predictor import gt_columns


def test_eval_label_matrix(golden):
    g = golden('eval')
    np.testing.assert_allclose(evaluation.eval(g['label/matrix']), g['label/metrics'], rtol=0, atol=1e-12)
    with pytest.raises(IndexError):
        evaluation.eval(np.zeros((2, 3)))


def test_eval_qry2retro(golden):
    g = golden('eval')
    np.testing.assert_allclose(evaluation.eval_qry2retro(g['q2r/S'], 1), g['q2r/metrics'], rtol=0, atol=1e-12)


def test_eval_via_label_matrix_matches_predictor_fixture(golden):
    """Build the label matrix the way predictor.py does (argsort) and feed our eval: same 7 numbers."""
    g = golden('eval')
    for c in g.json('cases'):
        k = c['key']
        S, gt = g[k + '/S'], g[k + '/gt']
        inds = np.argsort(S, axis=1)[:, ::-1]
        lab = (inds == gt[:, None]).astype(float)
        np.testing.assert_allclose(evaluation.eval(lab), g[k + '/t2v'], rtol=0, atol=1e-12)


def test_np_l2norm(golden):
    g = golden('txt2vis')
    np.testing.assert_allclose(evaluation.l2norm(g['l2/x']).astype(np.float32), g['l2/np'], rtol=0, atol=1e-7)


def test_bigfile(golden):
    exp = json.load(open(os.path.join(GOLDEN, 'bigfile_expect.json')))
    for name, e in exp.items():
        bf = BigFile(os.path.join(GOLDEN, name))
        assert bf.shape() == e['shape'] and bf.names == e['names']
        assert bf.ndims == e['ndims'] and bf.nr_of_images == e['nr_of_images']
        names, vecs = bf.read(e['request'])
        assert names == e['read_names']
        np.testing.assert_array_equal(np.array(vecs, np.float32), np.array(e['read_vecs'], np.float32))
        assert [list(x) for x in bf.read(['zzz'])] == e['read_empty']
        names, vecs = bf.read([3, 1], isname=False)
        assert names == e['by_index_names']
        np.testing.assert_array_equal(np.array(vecs, np.float32), np.array(e['by_index_vecs'], np.float32))
        np.testing.assert_array_equal(np.array(bf.read_one(e['names'][1]), np.float32), np.array(e['read_one'], np.float32))
        with pytest.raises(IndexError):
            bf.read_one('missing')
        assert e['read_one_missing_error'] == 'IndexError'
        m = bf.read_matrix([e['names'][2], e['names'][0], e['names'][2]])
        np.testing.assert_array_equal(m, np.array(e['matrix'], np.float32)[[2, 0, 2]])


def test_gt_columns():
    owner = gt_columns(['b#0', 'a#1', 'b#enc#2'.replace('#enc', '')], ['a', 'b'])
    assert owner.tolist() == [1, 0, 1]
    with pytest.raises(IndexError):
        gt_columns(['zz#0'], ['a'])
    with pytest.raises(ValueError):
        gt_columns(['a#0'], ['a', 'a'])


def _model_keys(model):
    return {k for k in model.state_dict().keys() if not k.startswith('txt_net.encoder.')}


def test_state_dict_keys_match_reference_laff(golden):
    g = golden('laff_towers')
    for c in g.json('cases'):
        cfg = make_config(c['vid_dims'], c['txt_dims'], c['D'], c['H'], 'LAFF', c['vis_no_transform'],
                          c['txt_no_transform'], with_ave=c['with_ave'], mul=c['mul'], batch_norm=c['batch_norm'])
        model = get_model('LAFF', 'cpu', cfg)
        ref = {k for k in g.sub(c['key'] + '/sd/').keys()}
        assert _model_keys(model) == ref
        sd = g.sub(c['key'] + '/sd/')
        for k, v in model.state_dict().items():
            if k in sd:
                assert tuple(v.shape) == tuple(sd[k].shape), k


def test_state_dict_keys_match_reference_framelaff(golden):
    g = golden('framelaff')
    for c in g.json('cases'):
        cfg = make_config(c['vid_dims'], {'bow': 20, 'CLIP': 512}, c['D'], c['H'], 'FrameLAFF',
                          vis_no_transform=c['frame_feats'], txt_no_transform=['CLIP_encoder'],
                          frame_feats={f: 512 for f in c['frame_feats']}, batch_norm=c['batch_norm'],
                          vis_frame_attention=c['vis_frame_attention'], vis_frame_addFC=c['vis_frame_addFC'],
                          frame_feat_with_video_feat=c['frame_feat_with_video_feat'])
        model = get_model('FrameLAFF', 'cpu', cfg)
        ref = set(g.sub(c['key'] + '/sd/').keys())
        ours = {k for k in _model_keys(model) if k.startswith('vis_net.')}
        assert ours == ref


def test_training_mode_and_out_of_scope_models_fail_loudly():
    cfg = make_config({'a': 16}, {'bow': 8}, 64, 1, txt_attention='attention_noAveNoAverageMul',
                      vis_attention='attention_noAveNoAverageMul')
    m = get_model('w2vpp_mutivis_attention', 'cpu', cfg)
    with pytest.raises(NotImplementedError):
        m(None)
    with pytest.raises(NotImplementedError):
        get_model('End2EndClip', 'cpu', cfg)
    m.train()
    with pytest.raises(NotImplementedError):
        m.vis_net({'a': torch.zeros(2, 16)})


def test_spec_workload_features_and_csr_row_shards():
    """C1-shaped synthetic features on the CPU: bow is CSR with int32 indices, 20 captions per video, shards re-assemble."""
    import torch
    from laff_amd import synth
    spec = synth.SPECS['tiny_c1']
    Nt, Nv = 600, 30
    vis, txt, gt, _ = synth.make_spec_features(spec, Nt, Nv, torch.device('cpu'))
    bow = txt['bow_encoding']
    assert bow.layout == torch.sparse_csr and bow.crow_indices().dtype == torch.int32 and bow.shape == (Nt, spec['txt']['bow'])
    assert vis['clip_ft'].shape == (Nv, 64) and vis['x3d'].shape == (Nv, 96) and txt['CLIP_encoding'].shape == (Nt, 64)
    assert torch.equal(gt, (torch.arange(Nt) // 20).to(torch.int32))
    dense = bow.to_dense()
    assert dense.min() >= 0 and (dense.sum(1) >= 1).all()
    parts = [synth.slice_rows(bow, lo, hi).to_dense() for lo, hi in ((0, 1), (1, 250), (250, 600))]
    assert torch.equal(torch.cat(parts), dense)
    assert torch.equal(synth.slice_rows(vis['x3d'], 3, 9), vis['x3d'][3:9])


def test_txt2vec_host_side_matches_reference_fixture():
    """laff_amd.txt2vec: tokenisation, id mapping and CSR construction (host logic; the gather-sum itself is a GPU test)."""
    import json
    import os
    import pickle
    import tempfile
    import torch
    from laff_amd import txt2vec as T
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'txt2vec.npz'))
    caps = json.loads(str(z['captions']))
    stop = set(json.loads(str(z['stopwords'])))
    vocab = json.loads(str(z['vocab']))
    assert [T.tokenize(c) for c in caps] == json.loads(str(z['tokens_all']))
    assert [T.tokenize(c, True, True, stop) for c in caps] == json.loads(str(z['tokens_nsw']))
    for key, sw in (('bow', None), ('bow_nsw', stop)):
        t2v = T.BowVec(vocab, sw)
        assert t2v.ndims == len(vocab) == len(t2v)
        assert np.array_equal(np.stack([t2v.encoding(c) for c in caps]), z[key])
        csr = t2v.csr(caps, torch.device('cpu'))
        assert csr.layout == torch.sparse_csr and csr.crow_indices().dtype == torch.int32
        assert np.array_equal(csr.to_dense().numpy(), z[key].astype(np.float32))
    words = json.loads(str(z['w2v_words']))
    for key, sw in (('w2v', None), ('w2v_nsw', stop)):
        t2v = T.W2Vec(words, z['w2v_table'], sw)
        assert np.allclose(np.stack([t2v.encoding(c) for c in caps]), z[key], rtol=0, atol=1e-15)
        dense = t2v.csr(caps, torch.device('cpu')).to_dense().numpy().astype(np.float64) @ z['w2v_table'].astype(np.float64)
        assert np.abs(dense - z[key]).max() <= 1e-6
    # a vocabulary pickled under a module that is not importable here (the reference pickles textlib.Vocabulary) loads into ours
    import sys
    import types
    mod = types.ModuleType('textlib_standin')

    class Vocabulary(object):
        pass
    Vocabulary.__module__, Vocabulary.__qualname__ = 'textlib_standin', 'Vocabulary'
    mod.Vocabulary = Vocabulary
    sys.modules['textlib_standin'] = mod
    v = Vocabulary()
    v.word2idx = {w: i for i, w in enumerate(vocab)}
    v.idx2word = {i: w for i, w in enumerate(vocab)}
    v.encoding = 'bow'
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'bow_nsw_5.pkl')
        pickle.dump(v, open(path, 'wb'))
        del sys.modules['textlib_standin']
        t2v = T.BowVec(path, stop)
        assert isinstance(t2v.vocab, T.Vocabulary) and t2v.vocab.find(vocab[3]) == 3 and t2v.ndims == len(vocab)


def test_gw_linear_decay_per_epoch():
    """W2VVPP_MutiVis.change_raw_global_emb_weight (model/model.py:1910-1941): gw <- max(0, gw + rate - 1) on both towers."""
    import torch
    from laff_amd.config import make_config
    from laff_amd.model import get_model
    cfg = make_config({'a': 8, 'b': 8}, {'bow': 8, 'w2v': 8}, 16, 2, 'LAFF', with_ave=True)
    cfg.txt_attention_global_decay_rate, cfg.vis_attention_global_decay_rate = 0.8, 0.7
    model = get_model('LAFF', torch.device('cpu'), cfg)
    t, v = model.txt_net.attention_layer, model.vis_net.attention_layer
    assert t.get_raw_global_emb_weight() == 1.0 and v.get_raw_global_emb_weight() == 1.0
    seen = []
    for _ in range(6):
        model.change_raw_global_emb_weight()
        seen.append((round(t.get_raw_global_emb_weight(), 6), round(v.get_raw_global_emb_weight(), 6)))
    assert seen == [(0.8, 0.7), (0.6, 0.4), (0.4, 0.1), (0.2, 0.0), (0.0, 0.0), (0.0, 0.0)]
    for layer in (t, v):                                      # every head carries the same value
        sd = layer.state_dict()
        vals = [float(x) for k, x in sd.items() if k.endswith('global_emb_weight_net.weight')]
        assert len(vals) == 2 and len(set(vals)) == 1


def test_evaluation_l2norm_golden(golden):
    g = golden('eval_cosine')
    np.testing.assert_allclose(evaluation.l2norm(g['q']), g['l2q'], rtol=0, atol=1e-6)
