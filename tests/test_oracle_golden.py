"""Pins the CPU oracle (oracle/laff_oracle.py) against golden vectors produced by the real reference
(tools/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import laff_oracle as O

TOL = 2e-6   # fp32 outputs are unit-norm vectors / cosines in [-1, 1]


def close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
    assert d <= tol, d


def test_attention_1(golden):
    g = golden('attention_1')
    for c in g.json('cases'):
        k = c['key']
        x = g[k + '/x'] if c.get('own_x') else g['x']
        out, a = O.attention_1(x, g[k + '/w'], float(g[k + '/b']), c['with_ave'], c['mul'], c['gw'], True)
        close(out, g[k + '/out'])
        if k + '/weights' in g:
            exp = g[k + '/weights']
            if c['with_ave']:   # the reference stashes weights + gw/L in that case (Attention.py:97)
                a = a + np.float32(c['gw']) / a.shape[1]
            close(a, exp)
    close(O.just_average(g['x']), g['just_average/out'])


def test_multi_head(golden):
    g = golden('multi_head')
    for c in g.json('cases'):
        k = c['key']
        sd = g.sub(k + '/sd/')
        att = O.attention_from_sd(sd, '', c['H'], c['with_ave'], c['mul'], c['split_head'], c['l2norm_each_head'])
        out = O.multi_head_attention(g[k + '/x'], att['w'], att['b'], att['gw'], c['H'], c['with_ave'], c['mul'],
                                     c['split_head'], c['l2norm_each_head'])
        close(out, g[k + '/out'])


def test_transform_net(golden):
    g = golden('transform_net')
    for c in g.json('cases'):
        k = c['key']
        sd = g.sub(k + '/sd/')
        spec = O.feature_spec(sd, '', g[k + '/x'], c['activation'], 1, not c['fc'])
        y = O.transform_net(**spec)
        close(y, g[k + '/y'], 2e-5)   # un-normalised activations up to |y| ~ 5 after BN


def _vis_specs(g, k, c, sd, prefix):
    specs = []
    for name in c['vid_dims']:
        nt = name in c['vis_no_transform']
        specs.append(O.feature_spec(sd, prefix + name + '.', g[k + '/vis/' + name], 'tanh', c['H'], nt))
    return specs


TXT_KEY = {'bow_encoder': 'bow_feature', 'w2v_encoder': 'w2v_feature', 'CLIP_encoder': 'CLIP_encoding'}


def _txt_specs(g, k, c, sd, key_prefix='/txt/'):
    specs = []
    for enc in c['encoder_name_list']:
        nt = enc in c['txt_no_transform']
        specs.append(O.feature_spec(sd, 'txt_net.transform_layer.%s_transform.' % enc, g[k + key_prefix + TXT_KEY[enc]],
                                    'tanh', c['H'], nt))
    return specs


def test_laff_towers(golden):
    g = golden('laff_towers')
    for c in g.json('cases'):
        k = c['key']
        sd = g.sub(k + '/sd/')
        att_v = O.attention_from_sd(sd, 'vis_net.attention_layer.', c['H'], c['with_ave'], c['mul'])
        att_t = O.attention_from_sd(sd, 'txt_net.attention_layer.', c['H'], c['with_ave'], c['mul'])
        ve = O.fuse_tower(_vis_specs(g, k, c, sd, 'vis_net.VisMutiTransformNet.'), att_v, c['H'])
        te = O.fuse_tower(_txt_specs(g, k, c, sd), att_t, c['H'])
        close(ve, g[k + '/vis_emb'])
        close(te, g[k + '/txt_emb'])
        close(O.txt2vis_matrix(te, ve), g[k + '/scores'])
        close(O.txt2vis_matrix_fast(te, ve), g[k + '/scores'])


def test_laff_expert_embedding(golden):
    """expert embeddings added to the stacked planes, with and without l2norm(dim=2) (model/model.py:1866-1873, :1686-1694)"""
    g = golden('laff_expert')
    for c in g.json('cases'):
        k = c['key']
        sd = g.sub(k + '/sd/')
        att_v = O.attention_from_sd(sd, 'vis_net.attention_layer.', c['H'], c['with_ave'], c['mul'])
        att_t = O.attention_from_sd(sd, 'txt_net.attention_layer.', c['H'], c['with_ave'], c['mul'])
        ve = O.fuse_tower(_vis_specs(g, k, c, sd, 'vis_net.VisMutiTransformNet.'), att_v, c['H'],
                          sd['vis_net.expert_embedding.weight'], c['l2norm'])
        te = O.fuse_tower(_txt_specs(g, k, c, sd), att_t, c['H'], sd['txt_net.expert_embedding.weight'], c['l2norm'])
        close(ve, g[k + '/vis_emb'])
        close(te, g[k + '/txt_emb'])


def test_framelaff(golden):
    g = golden('framelaff')
    for c in g.json('cases'):
        k = c['key']
        sd = g.sub(k + '/sd/')
        H = c['H']
        with_ave, mul = O.FRAME_ATTENTION_FLAGS[c['vis_frame_attention']]
        f = c['frame_feats'][0]
        ai = 1 if c['vis_frame_addFC'] else 0
        p = 'vis_net.frame_attention.%s.%d.' % (f, ai)
        Wfc = sd.get('vis_net.frame_attention.%s.0.weight' % f) if c['vis_frame_addFC'] else None
        bfc = sd.get('vis_net.frame_attention.%s.0.bias' % f) if c['vis_frame_addFC'] else None
        fv = O.frame_attention(g[k + '/frames'], sd[p + 'embedding_common.0.weight'].reshape(-1),
                               float(sd[p + 'embedding_common.0.bias'].reshape(())), with_ave, mul,
                               float(sd[p + 'global_emb_weight_net.weight'].reshape(())), Wfc, bfc)
        if k + '/frame_vec' in g:
            close(fv, g[k + '/frame_vec'])
        specs = []
        if c['frame_feat_with_video_feat']:
            for name in c['vid_dims']:
                specs.append(O.feature_spec(sd, 'vis_net.%s.' % name, g[k + '/vis/' + name], 'tanh', H, False))
        specs.append(O.feature_spec(sd, 'vis_net.%s.' % f, fv, None, H, True))
        att = O.attention_from_sd(sd, 'vis_net.vis_attention_layer.', H, False, False)
        close(O.fuse_tower(specs, att, H), g[k + '/vis_emb'])


def test_framelaff_masking_equivalence(golden):
    """Without a frame FC, dropping the zero-padded frames changes nothing after the eps=0 L2 norm
     when with_ave is off and mul is off."""
    g = golden('framelaff')
    c = g.json('cases')[0]
    k = c['key']
    sd = g.sub(k + '/sd/')
    p = 'vis_net.frame_attention.%s.0.' % c['frame_feats'][0]
    w = sd[p + 'embedding_common.0.weight'].reshape(-1)
    b = float(sd[p + 'embedding_common.0.bias'].reshape(()))
    frames, lens = g[k + '/frames'], g[k + '/lens']
    for i in range(frames.shape[0]):
        un = O.attention_1(frames[i:i + 1, :lens[i]], w, b)
        close(un[0], g[k + '/frame_vec'][i], 5e-7)


def test_txt2vis(golden):
    g = golden('txt2vis')
    close(O.txt2vis_matrix(g['t2'], g['v2']), g['s2'])
    close(O.txt2vis_matrix(g['t3'], g['v3']), g['s3'])
    close(O.txt2vis_matrix_fast(g['t3'], g['v3']), g['s3'])
    close(O.txt2vis_matrix(g['t3u'], g['v3u']), g['s3u'])
    close(O.l2norm(g['l2/x']), g['l2/default'])
    close(O.l2norm(g['l2/x'], eps=0.0)[[0, 1, 3, 4, 5]], g['l2/eps0'][[0, 1, 3, 4, 5]])
    close(O.np_l2norm(g['l2/x']).astype(np.float32), g['l2/np'])
    close(O.np_cosine_sim(g['t2'], g['v2']).astype(np.float32), g['cos/np'])


def test_predict(golden):
    g = golden('predict')
    c = g.json('cfg')
    sd = g.sub('sd/')
    H, bs = c['H'], c['bs']
    perm = g['perm']
    vis_specs = [O.feature_spec(sd, 'vis_net.VisMutiTransformNet.%s.' % n, g['vis/' + n], 'tanh', H,
                                n in c['vis_no_transform']) for n in c['vid_dims']]
    att_v = O.attention_from_sd(sd, 'vis_net.attention_layer.', H, False, False)
    att_t = O.attention_from_sd(sd, 'txt_net.attention_layer.', H, False, False)
    ve = O.fuse_tower(vis_specs, att_v, H)
    close(ve, g['video_all_embs'])
    txt_specs = [O.feature_spec(sd, 'txt_net.transform_layer.%s_transform.' % e, g['txt/' + TXT_KEY[e]][perm], 'tanh', H,
                                e in c['txt_no_transform']) for e in c['encoder_name_list']]
    te = O.fuse_tower(txt_specs, att_t, H)
    Nt, Nv = te.shape[0], ve.shape[0]
    tb = [(np.arange(s, min(Nt, s + bs)), te[s:s + bs]) for s in range(0, Nt, bs)]
    vb = [(np.arange(s, min(Nv, s + bs)), ve[s:s + bs]) for s in range(0, Nv, bs)]
    scores = O.predict_blocked(tb, vb, Nt, Nv)
    close(scores, g['scores'])
    txt_ids = [g.json('txt_ids')[i] for i in perm]
    assert txt_ids == g.json('txt_ids_out')
    t2v, v2t = O.predictor_metrics(scores, txt_ids, g.json('vis_ids'))
    np.testing.assert_allclose(t2v, g['t2v_metrics'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(v2t, g['v2t_metrics'], rtol=0, atol=1e-12)


def test_eval(golden):
    g = golden('eval')
    for c in g.json('cases'):
        k = c['key']
        t2v, v2t = O.predictor_metrics(g[k + '/S'], g.json(k + '/txt_ids'), g.json(k + '/vis_ids'))
        np.testing.assert_allclose(t2v, g[k + '/t2v'], rtol=0, atol=1e-12)
        np.testing.assert_allclose(v2t, g[k + '/v2t'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.eval_label_matrix(g['label/matrix']), g['label/metrics'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.eval_qry2retro(g['q2r/S'], 1), g['q2r/metrics'], rtol=0, atol=1e-12)


def test_bigfile_fixture_is_wellformed():
    from conftest import GOLDEN
    exp = json.load(open(os.path.join(GOLDEN, 'bigfile_expect.json')))
    for name, e in exp.items():
        mat = np.fromfile(os.path.join(GOLDEN, name, 'feature.bin'), dtype=np.float32).reshape(e['shape'])
        np.testing.assert_array_equal(mat, np.array(e['matrix'], np.float32))


def test_txt2vec_oracle_matches_reference_encoders(golden):
    """tokenizer + bow count vectors + w2v mean-pool against txt2vec.BowVec(NSW) / W2Vec(NSW) outputs of the reference."""
    import json
    z = golden('txt2vec')
    caps = json.loads(str(z['captions']))
    stop = set(json.loads(str(z['stopwords'])))
    vocab = json.loads(str(z['vocab']))
    words = json.loads(str(z['w2v_words']))
    table = z['w2v_table']
    assert [O.tokenize(c) for c in caps] == json.loads(str(z['tokens_all']))
    assert [O.tokenize(c, True, True, stop) for c in caps] == json.loads(str(z['tokens_nsw']))
    assert [O.tokenize(c, clean=False) for c in caps] == json.loads(str(z['tokens_noclean']))
    for key, rm in (('bow', False), ('bow_nsw', True)):
        got = np.stack([O.bow_encoding(c, vocab, rm, stop) for c in caps])
        assert np.array_equal(got, z[key])
    for key, rm in (('w2v', False), ('w2v_nsw', True)):
        got = np.stack([O.w2v_encoding(c, words, table, rm, stop) for c in caps])
        assert np.array_equal(got, z[key])


def test_margin_ranking_loss_oracle_matches_reference_autograd(golden):
    g = golden('margin_loss')
    for c in g.json('cases'):
        k = c['key']
        loss, d_s, d_im = O.margin_ranking_loss(g[k + '/s'], g[k + '/im'], c['margin'], c['max_violation'], c['cost_style'],
                                                c['direction'])
        assert abs(float(loss) - float(g[k + '/loss'])) <= 2e-5 * max(1.0, abs(float(g[k + '/loss']))), c
        assert np.abs(d_s - g[k + '/d_s']).max() <= 2e-6, c
        assert np.abs(d_im - g[k + '/d_im']).max() <= 2e-6, c
        assert float(g[k + '/loss']) > 0 and np.abs(g[k + '/d_s']).max() > 0      # the fixture exercises violations


def test_eval_cosine_numpy_api(golden):
    """evaluation.l2norm / evaluation.cosine_sim (evaluation.py:11-16, 44-49), incl. a zero and a near-epsilon query row"""
    g = golden('eval_cosine')
    close(O.np_l2norm(g['q']), g['l2q'], 1e-6)
    close(O.np_cosine_sim(g['q'], g['r']), g['sim'], 1e-6)
