"""Model-level parity on a real MI355X: laff_amd towers / predict / metrics against the reference's golden outputs."""
import numpy as np
import pytest
import torch

from oracle import laff_oracle as O
from laff_amd.config import make_config
from laff_amd.model import get_model
from util import load_sd, maxdiff

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TXT_KEY = {'bow_feature': 'bow_encoding', 'w2v_feature': 'w2v_encoding', 'CLIP_encoding': 'CLIP_encoding'}


def t(a):
    return torch.from_numpy(np.array(a))     # CPU tensors: the towers move them, like the reference


def test_laff_towers_golden(golden):
    g = golden('laff_towers')
    for c in g.json('cases'):
        k = c['key']
        cfg = make_config(c['vid_dims'], c['txt_dims'], c['D'], c['H'], 'LAFF', c['vis_no_transform'],
                          c['txt_no_transform'], with_ave=c['with_ave'], mul=c['mul'], batch_norm=c['batch_norm'])
        model = get_model('LAFF', DEV, cfg).eval()
        res = load_sd(model, g.sub(k + '/sd/'))
        assert not res.unexpected_keys and not res.missing_keys
        vis_in = {n: t(g[k + '/vis/' + n]) for n in c['vid_dims']}
        cap = {'caption': ['x'] * 24}
        cap.update({TXT_KEY[n]: t(v) for n, v in g.sub(k + '/txt/').items()})
        ve = model.vis_net(vis_in)
        te = model.txt_net(cap)
        assert tuple(ve.shape) == (24, c['H'], c['D'] // c['H'])
        assert maxdiff(ve, g[k + '/vis_emb']) <= 5e-6
        assert maxdiff(te, g[k + '/txt_emb']) <= 5e-6
        assert maxdiff(model.get_txt2vis_matrix(te, ve, precision='fp16x3'), g[k + '/scores']) <= 5e-6
        assert maxdiff(model.get_txt2vis_matrix(te, ve), g[k + '/scores']) <= 1e-4     # default fp16 operands
        if 'bow_encoding' in cap:
            # the same bag-of-words feature handed over as a CSR matrix: gathered inside the fuse launch (gather plane)
            dense = cap['bow_encoding'].to(DEV).float()
            sp = dense.to_sparse_csr()
            cap_sp = dict(cap, bow_encoding=torch.sparse_csr_tensor(sp.crow_indices().to(torch.int32), sp.col_indices().to(torch.int32),
                                                                     sp.values(), size=sp.shape))
            assert maxdiff(model.txt_net(cap_sp), g[k + '/txt_emb']) <= 5e-6
        assert vis_in['X3D_L'].is_cuda       # in-place device move of the caller's dict, like the reference
        assert maxdiff(model.encode_video({n: t(g[k + '/vis/' + n]) for n in c['vid_dims']}), g[k + '/vis_emb']) <= 5e-6


def test_laff_expert_embedding_golden(golden):
    """vis/txt_expert_embedding {'expert': True, 'l2norm': False | True}: the embedding row rides in the plane's shift, the row norm
    over all heads is one extra launch (laff_plane_row_norms) -- against the reference's towers."""
    g = golden('laff_expert')
    for c in g.json('cases'):
        k = c['key']
        cfg = make_config(c['vid_dims'], c['txt_dims'], c['D'], c['H'], 'LAFF', c['vis_no_transform'], c['txt_no_transform'],
                          with_ave=c['with_ave'], mul=c['mul'], batch_norm=c['batch_norm'],
                          vis_expert_embedding={'expert': True, 'l2norm': c['l2norm']},
                          txt_expert_embedding={'expert': True, 'l2norm': c['l2norm']})
        model = get_model('LAFF', DEV, cfg).eval()
        res = load_sd(model, g.sub(k + '/sd/'))
        assert not res.unexpected_keys and not res.missing_keys
        N = g[k + '/vis_emb'].shape[0]
        vis_in = {n: t(g[k + '/vis/' + n]) for n in c['vid_dims']}
        cap = {'caption': ['x'] * N}
        cap.update({TXT_KEY[n]: t(v) for n, v in g.sub(k + '/txt/').items()})
        assert maxdiff(model.vis_net(vis_in), g[k + '/vis_emb']) <= 5e-6, c
        assert maxdiff(model.txt_net(cap), g[k + '/txt_emb']) <= 5e-6, c
        # the sparse bag-of-words input takes the same route (projected first when a row norm is needed)
        sp = cap['bow_encoding'].to(DEV).float().to_sparse_csr()
        cap_sp = dict(cap, bow_encoding=torch.sparse_csr_tensor(sp.crow_indices().to(torch.int32), sp.col_indices().to(torch.int32),
                                                                 sp.values(), size=sp.shape))
        assert maxdiff(model.txt_net(cap_sp), g[k + '/txt_emb']) <= 5e-6, c


def test_attention_1_family_with_a_repeated_no_transform_feature():
    """vis_attention of the single-vector Attention_1 family with heads > 1 and a no-transform feature: the reference repeats the
    feature `heads` times whatever the attention type (model/model.py:1822-1823); the tiled plane is written out to full width."""
    from laff_amd import ops
    from laff_amd.model.Attention import Attention_1
    g = np.random.default_rng(5)
    N, H, d = 33, 4, 64
    D = H * d
    x_raw = g.normal(0, 1, (N, d)).astype(np.float32)
    x_fc = g.normal(0, 1, (N, D)).astype(np.float32)
    scale, shift = g.uniform(0.5, 1.5, D).astype(np.float32), g.normal(0, 0.1, D).astype(np.float32)
    att = Attention_1(D, with_ave=False, mul=False).to(DEV).eval()
    dv = lambda a: torch.from_numpy(a).to(DEV)
    out = att.fuse_planes([(dv(x_raw), True, dv(scale), dv(shift)), (dv(x_fc), False, None, None)], heads=H)
    tiled = np.tile(x_raw, (1, H)) * scale + shift
    w = att.embedding_common[0].weight.detach().cpu().numpy().reshape(-1)
    b = att.embedding_common[0].bias.detach().cpu().numpy().reshape(())
    ref = O.attention_1(np.stack([tiled, x_fc], axis=1), w, b, False, False, 1.0)
    assert maxdiff(out, ref) <= 3e-6


def test_framelaff_golden(golden):
    g = golden('framelaff')
    for c in g.json('cases'):
        k = c['key']
        cfg = make_config(c['vid_dims'], {'bow': 20, 'CLIP': 512}, c['D'], c['H'], 'FrameLAFF',
                          vis_no_transform=c['frame_feats'], txt_no_transform=['CLIP_encoder'],
                          frame_feats={f: 512 for f in c['frame_feats']}, batch_norm=c['batch_norm'],
                          vis_frame_attention=c['vis_frame_attention'], vis_frame_addFC=c['vis_frame_addFC'],
                          frame_feat_with_video_feat=c['frame_feat_with_video_feat'])
        model = get_model('FrameLAFF', DEV, cfg).eval()
        res = load_sd(model, g.sub(k + '/sd/'))
        assert not res.unexpected_keys
        vis_in = {n: t(g[k + '/vis/' + n]) for n in c['vid_dims']}
        frame_in = {'mask_tensor': t(g[k + '/mask']), c['frame_feats'][0]: t(g[k + '/frames'])}
        ve = model.vis_net(vis_in, vis_frame_feat_dict_input=frame_in)
        assert maxdiff(ve, g[k + '/vis_emb']) <= 5e-6, c
        if k + '/frame_vec' in g:
            assert maxdiff(vis_in[c['frame_feats'][0]], g[k + '/frame_vec']) <= 5e-6


class _DS:
    def __init__(self, n):
        self.length = n

    def __len__(self):
        return self.length


class VisLoader:
    def __init__(self, feats, ids, bs):
        self.feats, self.ids, self.batch_size, self.dataset = feats, ids, bs, _DS(len(ids))

    def __len__(self):
        return (len(self.ids) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for s in range(0, len(self.ids), self.batch_size):
            e = min(len(self.ids), s + self.batch_size)
            yield {'vis_feat_dict': {k: t(v[s:e]) for k, v in self.feats.items()}, 'idxs': list(range(s, e)),
                   'vis_ids': tuple(self.ids[s:e]), 'vis_frame_feat_dict': {}, 'vis_origin_frame_tuple': (None,) * (e - s)}


class TxtLoader:
    def __init__(self, feats, ids, bs, perm):
        self.feats, self.ids, self.batch_size, self.perm, self.dataset = feats, ids, bs, perm, _DS(len(ids))

    def __len__(self):
        return (len(self.ids) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for s in range(0, len(self.ids), self.batch_size):
            order = self.perm[s:min(len(self.ids), s + self.batch_size)]
            cap = {'caption': [self.ids[i] for i in order]}
            cap.update({k: t(v[order]) for k, v in self.feats.items()})
            yield cap, [int(i) for i in order], tuple(self.ids[i] for i in order)


def test_predict_golden(golden):
    """predict() end to end: same (scores, txt_ids, vis_ids) triple and the same 7+7 metrics as the reference."""
    from laff_amd import predictor
    g = golden('predict')
    c = g.json('cfg')
    cfg = make_config(c['vid_dims'], c['txt_dims'], c['D'], c['H'], 'LAFF', c['vis_no_transform'], c['txt_no_transform'])
    model = get_model('LAFF', DEV, cfg).eval()
    res = load_sd(model, g.sub('sd/'))
    assert not res.unexpected_keys and not res.missing_keys
    vis_ids, txt_ids = g.json('vis_ids'), g.json('txt_ids')
    vl = VisLoader({n: g['vis/' + n] for n in c['vid_dims']}, vis_ids, c['bs'])
    tl = TxtLoader({TXT_KEY[k]: v for k, v in g.sub('txt/').items()}, txt_ids, c['bs'], g['perm'])
    model.sim_precision = 'fp16x3'
    scores, out_txt, out_vis = model.predict(tl, vl, 'cosine', record_emb=True)
    assert isinstance(scores, np.ndarray) and scores.dtype == np.float32
    assert list(out_txt) == g.json('txt_ids_out') and list(out_vis) == g.json('vis_ids_out')
    assert maxdiff(scores, g['scores']) <= 5e-6
    assert maxdiff(model.video_all_embs, g['video_all_embs']) <= 5e-6
    # cached video embeddings are reused for the next query set (record_emb, model/model.py:1026-1034)
    before = model.video_all_embs.data_ptr()
    model.sim_precision = 'fp16'   # single-pass fp16 operands: scores inside the 1e-4 contract, text->video ranks still exact
    scores16, _, _ = model.predict(tl, vl, 'cosine', record_emb=True)
    assert model.video_all_embs.data_ptr() == before
    assert maxdiff(scores16, g['scores']) <= 1e-4
    model.sim_precision = None     # predict()'s own default: the split-product GEMM
    S, _, _ = model.retrieve(tl, vl, record_emb=True)
    assert maxdiff(S, g['scores']) <= 5e-6
    t2v, v2t = predictor.retrieval_metrics(S, out_txt, out_vis)
    np.testing.assert_allclose(t2v, g['t2v_metrics'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(v2t, g['v2t_metrics'], rtol=0, atol=1e-9)
    ranks_x3 = model.last_t2v_ranks.clone()
    assert torch.equal(predictor.t2v_ranks(S, predictor.gt_columns(out_txt, out_vis)), ranks_x3)    # S recounts to the exact ranks
    S16, _, _ = model.retrieve(tl, vl, record_emb=True, precision='fp16')
    t2v16, v2t16 = predictor.retrieval_metrics(S16, out_txt, out_vis)
    np.testing.assert_allclose(t2v16, g['t2v_metrics'], rtol=0, atol=1e-9)     # all seven: the ranks are exact whatever the operands
    assert torch.equal(model.last_t2v_ranks, ranks_x3)
    # ... and with the pipeline's state the video->text direction too (laff_v2t_count_exact): fp16 operands, the reference's 7 numbers
    t2v16e, v2t16e = predictor.retrieval_metrics(S16, out_txt, out_vis, state=model.last_rank_state)
    np.testing.assert_allclose(t2v16e, g['t2v_metrics'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(v2t16e, g['v2t_metrics'], rtol=0, atol=1e-9)
    heads, _, _ = model.predict_each_head(tl, vl, 'cosine')
    assert maxdiff(heads.mean(axis=0), g['scores']) <= 1e-4
    # the batches of a plain loader are coalesced into one launch set per tower; the per-batch route gives the same triple
    model.sim_precision = 'fp16x3'
    model.coalesce_loader_batches = False
    try:
        sb, tb, vb = model.predict(tl, vl, 'cosine')
    finally:
        model.coalesce_loader_batches = True
    sw, tw, vw = model.predict(tl, vl, 'cosine')
    assert list(tb) == list(tw) and list(vb) == list(vw)
    assert maxdiff(sb, sw) <= 2e-6 and maxdiff(sw, g['scores']) <= 5e-6


def test_single_head_model_runs_2d():
    """'w2vpp_mutivis_attention' with plain Attention_1 blocks: 2-D embeddings, d = D = 2048 (streaming fuse kernel)."""
    from oracle import laff_oracle as O
    cfg = make_config({'a': 40, 'b': 24}, {'bow': 30, 'w2v': 20}, 2048, 1, 'w2vpp_mutivis_attention',
                      txt_attention='attention_noAverageMul_Ave', vis_attention='average_AverageMul_noAve')
    torch.manual_seed(3)
    model = get_model('w2vpp_mutivis_attention', DEV, cfg).eval()
    model.vis_net.attention_layer.embedding_common[0].bias.data.fill_(0.2)
    model.txt_net.attention_layer.change_raw_global_emb_weight(0.6)
    g = np.random.default_rng(4)
    vis = {'a': g.normal(0, 1, (9, 40)).astype(np.float32), 'b': g.normal(0, 1, (9, 24)).astype(np.float32)}
    ve = model.vis_net({k: t(v) for k, v in vis.items()})
    assert tuple(ve.shape) == (9, 2048)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    specs = [O.feature_spec(sd, 'vis_net.VisMutiTransformNet.%s.' % n, vis[n], 'tanh', 1, False) for n in ('a', 'b')]
    att = dict(kind='attention_1', w=sd['vis_net.attention_layer.embedding_common.0.weight'].reshape(-1),
               b=float(sd['vis_net.attention_layer.embedding_common.0.bias'][0]), with_ave=False, mul=True, gw=1.0)
    assert maxdiff(ve, O.fuse_tower(specs, att, 1)) <= 5e-6
    txt = {'bow_encoding': g.normal(0, 1, (7, 30)).astype(np.float32), 'w2v_encoding': g.normal(0, 1, (7, 20)).astype(np.float32)}
    te = model.txt_net({'caption': [''] * 7, **{k: t(v) for k, v in txt.items()}})
    specs = [O.feature_spec(sd, 'txt_net.transform_layer.%s_transform.' % e, txt[e.replace('encoder', 'encoding')], 'tanh', 1, False)
             for e in ('bow_encoder', 'w2v_encoder')]
    att = dict(kind='attention_1', w=sd['txt_net.attention_layer.embedding_common.0.weight'].reshape(-1),
               b=float(sd['txt_net.attention_layer.embedding_common.0.bias'][0]), with_ave=True, mul=False, gw=0.6)
    assert maxdiff(te, O.fuse_tower(specs, att, 1)) <= 5e-6
    S = model.get_txt2vis_matrix(te, ve, precision='fp32')
    assert maxdiff(S, O.txt2vis_matrix(te.cpu().numpy(), ve.cpu().numpy())) <= 2e-6


def test_laff_towers_golden_fp16x3_fc(golden):
    """The towers with FC_PRECISION = 'fp16x3' stay inside the embedding tolerance of the fp32 path."""
    import laff_amd.model.model as M
    g = golden('laff_towers')
    M.FC_PRECISION = 'fp16x3'
    try:
        for c in g.json('cases'):
            k = c['key']
            cfg = make_config(c['vid_dims'], c['txt_dims'], c['D'], c['H'], 'LAFF', c['vis_no_transform'],
                              c['txt_no_transform'], with_ave=c['with_ave'], mul=c['mul'], batch_norm=c['batch_norm'])
            model = get_model('LAFF', DEV, cfg).eval()
            load_sd(model, g.sub(k + '/sd/'))
            ve = model.vis_net({n: t(g[k + '/vis/' + n]) for n in c['vid_dims']})
            cap = {'caption': ['x'] * 24}
            cap.update({TXT_KEY[n]: t(v) for n, v in g.sub(k + '/txt/').items()})
            te = model.txt_net(cap)
            assert maxdiff(ve, g[k + '/vis_emb']) <= 5e-6
            assert maxdiff(te, g[k + '/txt_emb']) <= 5e-6
    finally:
        M.FC_PRECISION = 'fp32'


def test_text_feature_producers_on_device_match_reference_fixture(golden):
    """bow: captions -> CSR -> gather-sum FC == the reference's dense count vector through nn.Linear; w2v: mean-pool."""
    import json
    from laff_amd import txt2vec as T
    from laff_amd.model.model import TransformNet
    z = golden('txt2vec')
    caps = json.loads(str(z['captions']))
    stop = set(json.loads(str(z['stopwords'])))
    vocab = json.loads(str(z['vocab']))
    words = json.loads(str(z['w2v_words']))
    w2v = T.W2Vec(words, z['w2v_table'], stop)
    got = T.W2VTxtEncoder(w2v, DEV)({'caption': caps})['text_features']
    assert got.shape == (len(caps), 20)
    assert np.abs(got.cpu().numpy() - z['w2v_nsw']).max() <= 1e-6
    bow = T.BowVec(vocab, stop)
    enc = T.BoWTxtEncoder(bow, DEV)
    x = enc({'caption': caps})['text_features']
    assert x.layout == torch.sparse_csr
    torch.manual_seed(5)
    import laff_amd.model.model as M
    M.device = torch.device(DEV)
    net = TransformNet((len(vocab), 64), None, None, True, 'tanh').to(DEV).eval()
    net.bn1.running_mean.normal_(0, 0.1)
    net.bn1.running_var.uniform_(0.5, 1.5)
    y = net(x)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    ref = O.transform_net(z['bow_nsw'].astype(np.float32), sd['fc1.weight'], sd['fc1.bias'], 'tanh',
                          O._bn_from_sd(sd, ''))
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-5


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the driver's keys plus roofline / cpu_baseline (tiny workload, child process)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--workload', 'tiny', '--steps', '4', '--warmup', '1'],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    assert line['n_gpus'] == 1 and line['steps'] == 4 and line['warmup'] == 1 and line['higher_is_better'] is True
    assert line['vs_baseline'] is None and line['data'] == 'synthetic' and 'workload' in line['config']
    assert line['value'] > 0 and abs(line['value'] - 512 * 192 / (line['ms_per_step'] * 1e-3)) <= 1e-6 * line['value']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in line['roofline'], k
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in line['cpu_baseline'], k
    # round 2: exact ranks reported against float64 scores, the count-only mode, the sustained loop, the CPU variants
    q = line['quality']['vs_strict_similarity']
    assert q['identical_ranks_frac'] == 1.0 and q['identical_ranks_frac_vs_fp64_scores'] == 1.0 and q['pair_list_overflow'] is False
    assert line['no_scores_mode']['metrics_equal_to_headline_mode'] is True and line['no_scores_mode']['roofline']['bound'] == 'mfma'
    assert line['sustained']['seconds'] >= 1.9
    assert {v['kind'] for v in line['cpu_baseline']['variants']} == {'port-blockloop-argsort', 'port-vectorised'}
    assert abs(line['cpu_baseline']['r1'] - line['quality']['R@1']) < 1e-9
    # round 5: per-kernel figures from stamps inside a capture of the timed step (every launch of the step listed, adding up to the
    # instrumented span), the strict mode, one rank's share of an 8-rank pass in both decompositions
    assert 'stamps' in line['kernels_source'] and line['stages_ms_eager_pass'] is None
    sb = line['step_breakdown']
    assert sb['launches'] == sum(k['launches_per_step'] for k in line['kernels'].values())
    assert abs(sb['kernels_ms'] + sb['gaps_ms'] + sb['launches'] * sb['stamp_cost_ms'] - sb['instrumented_span_ms']) <= 0.02 * sb['instrumented_span_ms'] + 1e-3
    assert {'fc_act_bn', 'fuse', 'sim_gemm', 'rank_resolve'} <= set(line['kernels'])
    assert line['roofline']['kernel'] is not None and line['roofline']['frac'] is not None
    assert line['strict_mode']['precision'] == 'fp16x3' and line['strict_mode']['metrics_equal_to_headline_mode'] is True
    emu = {e['scheme']: e for e in line['shard_emulation']}
    assert set(emu) == {'text', 'video'} and all(e['metrics_equal_to_unsharded'] and e['ms_per_step'] > 0 for e in emu.values())


def test_bench_emulate_shard_line():
    """bench.py --emulate-shard G: ONE line whose headline is rank 0's share of a G-rank pass (no collectives), ranks and metrics equal to
    the un-sharded pass (the bench raises otherwise)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for shard in ('text', 'video'):
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--workload', 'tiny', '--steps', '4', '--warmup', '1',
                              '--emulate-shard', '3', '--shard', shard], capture_output=True, text=True, timeout=900, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
        assert len(lines) == 1
        line = json.loads(lines[0])
        e = line['shard_emulation'][0]
        assert e['scheme'] == shard and e['ranks_of'] == 3 and e['metrics_equal_to_unsharded'] is True
        assert 'EMULATED' in line['config']['workload'] and line['n_gpus'] == 1 and line['value'] > 0
        assert e['score_block'] == ([171, 192] if shard == 'text' else [512, 64])


def test_evaluation_cosine_sim_golden(golden):
    """laff_amd.evaluation.cosine_sim (GPU-backed, numpy in / numpy out) against the reference's evaluation.cosine_sim
    (evaluation.py:44-49), every operand precision; a zero query row and one whose norm is comparable to the 1e-10 epsilon."""
    from laff_amd import evaluation
    g = golden('eval_cosine')
    for prec, tol in (('fp16x3', 2e-6), ('fp32', 2e-6), ('bf16x3', 5e-6), ('fp16', 4e-4)):     # d = 96: 16-bit rounding ~ 1/sqrt(d)
        got = evaluation.cosine_sim(g['q'], g['r'], precision=prec)
        assert isinstance(got, np.ndarray) and got.dtype == np.float32
        assert np.abs(got - g['sim']).max() <= tol, prec
        assert np.all(got[3] == 0.0)
    assert np.abs(evaluation.compute_sim(g['q'], g['r']) - g['sim']).max() <= 2e-6
    with pytest.raises(NotImplementedError):
        evaluation.compute_sim(g['q'], g['r'], measure='hist')


def test_oversized_fc_problem_is_cut_into_row_chunks():
    """FC_PRECISION 'fp16x3' on a feature matrix whose packed operand would pass the C ABI's 4 GiB limit: run_fc cuts it into row
    chunks (here the limit is lowered instead of allocating 4 GiB) -- same output as the uncut launch, preallocated `out` honoured."""
    import laff_amd.model.model as M
    from laff_amd import ops
    torch.manual_seed(3)
    x = torch.randn(5000, 256, device=DEV)
    W = torch.randn(192, 256, device=DEV) / 16
    b = torch.randn(192, device=DEV) * 0.1
    ws = ops.split_rows(W)
    ref = ops.fc_act_bn_split_grouped([dict(x=x, weight=W, weight_split=ws, bias=b, activation='tanh')])[0]
    out = torch.empty_like(ref)
    probs = [dict(x=x, weight=W, weight_split=ws, bias=b, activation='tanh', out=out),
             dict(x=x[:100], weight=W, weight_split=ws, bias=b, activation='tanh')]
    chunks = M._row_chunks(probs, limit_bytes=1 << 20)
    assert len(chunks) > 3 and sum(c['x'].shape[0] for c in chunks) == 5100
    M.FC_PRECISION = 'fp16x3'
    try:
        old = M._row_chunks
        M._row_chunks = lambda p: old(p, limit_bytes=1 << 20)
        got = M.run_fc(probs)
    finally:
        M.FC_PRECISION = 'fp32'
        M._row_chunks = old
    assert got[0] is out and torch.equal(out, ref) and torch.equal(got[1], ref[:100])


def test_retrieve_walks_the_overflow_ladder(monkeypatch):
    """model.retrieve() on degenerate scores (hundreds of near-duplicate videos inside every query's error band) with a pair list far
    too small: the default-size pass overflows, the 8x pass overflows, the pass on hi/lo split operands (band ~1e-6) fits -- nothing
    of an overflowed pass is published, and the ranks are those of the float64 scores (model/model.py:1018-1079 is the call this
    replaces; the ranking loop is predictor.py:232-244)."""
    from laff_amd import ops, predictor
    g = np.random.default_rng(7)
    cfg = make_config({'a': 64, 'b': 48}, {'bow': 40, 'w2v': 24}, 512, 1, 'LAFF', [], [])
    torch.manual_seed(11)
    model = get_model('LAFF', DEV, cfg).eval()
    Nv, per = 600, 3
    base_a, base_b = g.normal(0, 1, (1, 64)).astype(np.float32), g.normal(0, 1, (1, 48)).astype(np.float32)
    vis = {'a': base_a + 0.05 * g.normal(0, 1, (Nv, 64)).astype(np.float32),              # 600 videos clustered around one point
           'b': base_b + 0.05 * g.normal(0, 1, (Nv, 48)).astype(np.float32)}
    vis_ids = ['video%d' % i for i in range(Nv)]
    txt_ids = ['video%d#%d' % (i, k) for i in range(Nv) for k in range(per)]
    Nt = len(txt_ids)
    txt = {'bow_encoding': g.normal(0, 1, (Nt, 40)).astype(np.float32), 'w2v_encoding': g.normal(0, 1, (Nt, 24)).astype(np.float32)}
    vl = VisLoader(vis, vis_ids, 64)
    tl = TxtLoader(txt, txt_ids, 64, np.arange(Nt))
    calls = []
    real = ops.exact_ranks

    def spy(Et, Ev, T, V, gt, want_scores=True, col0=0, pair_cap=None):
        out = real(Et, Ev, T, V, gt, want_scores, col0, pair_cap)
        calls.append((T.precision, pair_cap, out[2].overflowed()))
        return out
    monkeypatch.setattr(ops, 'exact_ranks', spy)
    # retrieve() asks for the list sizes of the second and third attempt (8 x default) before the first attempt takes the default
    # itself: 8 slots, then 64, then 2^21 for the split operands
    sizes = iter([8, 1 << 18, 8])
    monkeypatch.setattr(ops, 'default_pair_cap', lambda n: next(sizes))
    S, out_txt, out_vis = model.retrieve(tl, vl, precision='fp16')
    assert [c[0] for c in calls] == ['fp16', 'fp16', 'fp16x3'] and [c[2] for c in calls] == [True, True, False]
    assert calls[0][1] is None and calls[1][1] == 64 and calls[2][1] == 1 << 21
    gt = predictor.gt_columns(out_txt, out_vis)
    te = torch.cat([model.txt_net(c) for c, _, _ in tl]).double().reshape(Nt, -1)
    ve = model.video_all_embs.double().reshape(Nv, -1)
    S64 = (te / (te.norm(dim=1, keepdim=True) + 1e-13)) @ (ve / (ve.norm(dim=1, keepdim=True) + 1e-13)).T
    want = 1 + ((S64 > S64[torch.arange(Nt), torch.as_tensor(gt, device=S64.device).long()][:, None]).sum(dim=1))
    assert torch.equal(model.last_t2v_ranks.long(), want)
    assert len(set(want.tolist())) > 50                       # degenerate, but the ranks are spread: equality is a real check
    # with the list it asks for from the start nothing is retried
    calls.clear()
    monkeypatch.setattr(ops, 'default_pair_cap', lambda n: 1 << 20)
    model.retrieve(tl, vl, precision='fp16')
    assert len(calls) == 1 and calls[0][2] is False
    assert torch.equal(model.last_t2v_ranks.long(), want)
