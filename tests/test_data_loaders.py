"""Bulk BigFile loaders (laff_amd/data.py): batch layout, frame grouping, and -- on a GPU -- predict() end to end from disk."""
import os

import numpy as np
import pytest
import torch

from laff_amd.bigfile import BigFile
from laff_amd.data import BulkTxtLoader, BulkVisLoader, write_bigfile


def _dataset(tmp_path, Nv=37, per=3, d=512, frames=0, seed=0):
    g = np.random.default_rng(seed)
    z = g.normal(0, 1, (Nv, 16)).astype(np.float32)
    vis_ids = ['video%d' % i for i in range(Nv)]
    feats = {}
    for name in ('clip_ft', 'x3d', 'ircsn', 'tf'):
        m = (z @ g.normal(0, 1, (16, d)) / 4 + 0.5 * g.normal(0, 1, (Nv, d))).astype(np.float32)
        order = g.permutation(Nv)                       # file order != request order
        write_bigfile(str(tmp_path / 'vis' / name), [vis_ids[i] for i in order], m[order])
        feats[name] = m
    caps, tfe = [], {}
    cap_ids = ['video%d#%d' % (v, k) for k in range(per) for v in range(Nv)]
    owner = np.array([int(c.split('#')[0][5:]) for c in cap_ids])
    for key in ('rnn_encoding', 'bow_encoding', 'w2v_encoding', 'CLIP_encoding'):
        m = (z[owner] @ g.normal(0, 1, (16, d)) / 4 + 0.5 * g.normal(0, 1, (len(cap_ids), d))).astype(np.float32)
        write_bigfile(str(tmp_path / 'txt' / key), cap_ids, m)
        tfe[key] = m
    capfile = str(tmp_path / 'caps.txt')
    with open(capfile, 'w') as f:
        for c in cap_ids:
            f.write('%s a caption for %s\n' % (c, c))
    return vis_ids, feats, cap_ids, tfe, capfile


def test_bulk_loaders_layout(tmp_path):
    vis_ids, feats, cap_ids, tfe, capfile = _dataset(tmp_path)
    vl = BulkVisLoader({n: BigFile(str(tmp_path / 'vis' / n)) for n in feats}, vis_ids, batch_size=16, device='cpu')
    assert len(vl) == 3 and vl.dataset.length == 37 and len(vl.dataset) == 37 and vl.batch_size == 16
    seen = []
    for b in vl:
        assert set(b) == {'vis_feat_dict', 'idxs', 'vis_ids', 'vis_frame_feat_dict', 'vis_origin_frame_tuple'}
        for n in feats:
            np.testing.assert_array_equal(b['vis_feat_dict'][n].numpy(), feats[n][b['idxs']])
        assert list(b['vis_ids']) == [vis_ids[i] for i in b['idxs']] and b['vis_frame_feat_dict'] == {}
        seen += b['idxs']
    assert seen == list(range(37))
    tl = BulkTxtLoader(capfile, {k: BigFile(str(tmp_path / 'txt' / k)) for k in tfe}, batch_size=50, device='cpu')
    assert len(tl) == 3 and len(tl.dataset) == len(cap_ids)
    rows = 0
    for cap, idxs, ids in tl:
        assert list(ids) == [cap_ids[i] for i in idxs] and cap['caption'][0].startswith('a caption for')
        for k in tfe:
            np.testing.assert_array_equal(cap[k].numpy(), tfe[k][idxs])
        rows += len(idxs)
    assert rows == len(cap_ids)
    assert tl.dataset.captions[cap_ids[5]] == 'a caption for ' + cap_ids[5]


def test_frame_grouping_and_padding(tmp_path):
    g = np.random.default_rng(1)
    vids = ['v_a', 'v_b', 'v_c']
    nfr = {'v_a': 5, 'v_b': 2, 'v_c': 7}
    fids, rows = [], []
    for v in vids:
        for k in g.permutation(nfr[v]):                  # frames stored out of order
            fids.append('%s_%d' % (v, k))
            rows.append(np.full(8, 100 * vids.index(v) + k, np.float32))
    write_bigfile(str(tmp_path / 'frames'), fids, np.stack(rows))
    vl = BulkVisLoader({}, vids, batch_size=3, device='cpu', vis_frame_feat_dicts={'ff': BigFile(str(tmp_path / 'frames'))},
                       max_frame=6)
    b = next(iter(vl))
    fr, mask = b['vis_frame_feat_dict']['ff'].numpy(), b['vis_frame_feat_dict']['mask_tensor'].numpy()
    assert fr.shape == (3, 6, 8) and mask.sum(axis=1).tolist() == [5, 2, 6]       # max_frame truncation (data_provider.py:476-477)
    assert fr[0, :5, 0].tolist() == [0, 1, 2, 3, 4] and fr[1, :2, 0].tolist() == [100, 101] and fr[1, 2:].sum() == 0
    assert fr[2, :, 0].tolist() == [200, 201, 202, 203, 204, 205]


@pytest.mark.gpu
def test_predict_from_disk_equals_device_pipeline(tmp_path):
    """BigFile directories -> bulk loaders -> model.predict() == the device-resident pipeline on the same matrices."""
    from laff_amd import predictor, retrieval, synth
    vis_ids, feats, cap_ids, tfe, capfile = _dataset(tmp_path, Nv=300, per=4)
    dev = torch.device('cuda')
    model = synth.build_model(1, 512, dev)
    vl = BulkVisLoader({n: BigFile(str(tmp_path / 'vis' / n)) for n in synth.VID_FEATS}, vis_ids, batch_size=128)
    tl = BulkTxtLoader(capfile, {k: BigFile(str(tmp_path / 'txt' / k)) for k in tfe}, batch_size=500)
    scores, txt_ids, out_vis = model.predict(tl, vl, 'cosine')
    assert list(txt_ids) == cap_ids and list(out_vis) == vis_ids and scores.shape == (1200, 300)
    owner = predictor.gt_columns(txt_ids, out_vis)
    res = retrieval.evaluate(model, {n: torch.from_numpy(feats[n]).to(dev) for n in synth.VID_FEATS},
                             {k: torch.from_numpy(v).to(dev) for k, v in tfe.items()},
                             torch.from_numpy(owner).to(dev), precision='fp16')
    assert float(np.abs(scores - res.S.cpu().numpy()).max()) <= 2e-4         # batched vs whole-matrix towers, fp16 GEMM
    t2v, v2t = predictor.retrieval_metrics(torch.from_numpy(scores).to(dev), txt_ids, out_vis)
    assert abs(t2v[0] - res.metrics[0]) <= 0.5 and t2v[3] == res.metrics[3]


class _NoWhole:
    """a loader without the whole() shortcut: retrieve() falls back to one tower pass per batch, like the reference"""

    def __init__(self, inner):
        self.inner, self.dataset, self.batch_size = inner, inner.dataset, inner.batch_size

    def __len__(self):
        return len(self.inner)

    def __iter__(self):
        return iter(self.inner)


@pytest.mark.gpu
def test_whole_matrix_route_equals_per_batch_route(tmp_path):
    """Bulk loaders hand retrieve() the whole matrices (one grouped FC launch, one fuse per side); the embeddings are bit-for-bit
    those of the per-batch route, the scores and the exact ranks are therefore identical too."""
    from laff_amd import predictor, synth
    vis_ids, feats, cap_ids, tfe, capfile = _dataset(tmp_path, Nv=300, per=4)
    dev = torch.device('cuda')
    model = synth.build_model(1, 512, dev)
    vl = BulkVisLoader({n: BigFile(str(tmp_path / 'vis' / n)) for n in synth.VID_FEATS}, vis_ids, batch_size=64)
    tl = BulkTxtLoader(capfile, {k: BigFile(str(tmp_path / 'txt' / k)) for k in tfe}, batch_size=64)
    whole = vl.whole()
    assert len(whole['vis_ids']) == 300 and whole['idxs'] == list(range(300))
    S_w, txt_w, vis_w = model.retrieve(tl, vl)
    emb_w, ranks_w = model.video_all_embs.clone(), model.last_t2v_ranks.clone()
    S_b, txt_b, vis_b = model.retrieve(_NoWhole(tl), _NoWhole(vl))
    assert list(txt_w) == list(txt_b) == cap_ids and list(vis_w) == list(vis_b) == vis_ids
    assert torch.equal(model.video_all_embs, emb_w)
    assert torch.equal(S_w, S_b) and torch.equal(model.last_t2v_ranks, ranks_w)
    assert torch.equal(predictor.t2v_ranks(S_w, predictor.gt_columns(txt_w, vis_w)), ranks_w)


def test_npy_features_stand_in_for_a_bigfile(tmp_path):
    """The numpy forms the reference reads (pickled {id: vector} dict, trainer.py:144-148; plain array + ids) behind the BigFile
    surface of the bulk loaders."""
    from laff_amd.data import BulkTxtLoader, NpyFeatures
    g = np.random.default_rng(3)
    ids = ['cap%d' % i for i in range(11)]
    mat = g.normal(0, 1, (11, 6)).astype(np.float32)
    path = str(tmp_path / 'feat.npy')
    np.save(path, {k: v for k, v in zip(ids, mat)}, allow_pickle=True)
    for src in (NpyFeatures(path), NpyFeatures(mat, ids)):
        assert src.shape() == [11, 6] and src.ndims == 6 and src.names == ids
        assert np.array_equal(src.read_matrix(['cap3', 'cap0', 'cap3']), mat[[3, 0, 3]])
        names, vecs = src.read(['cap9', 'nope', 'cap2', 'cap9'])
        assert names == ['cap2', 'cap9'] and np.allclose(vecs, mat[[2, 9]])
        assert np.allclose(src.read_one('cap5'), mat[5])
        with pytest.raises(IndexError):
            src.read_one('missing')
    loader = BulkTxtLoader([(i, 'a caption') for i in ids], {'CLIP_encoding': NpyFeatures(path)}, batch_size=4, device=None)
    got = np.concatenate([cap['CLIP_encoding'].numpy() for cap, _, _ in loader])
    assert np.array_equal(got, mat) and len(loader) == 3
