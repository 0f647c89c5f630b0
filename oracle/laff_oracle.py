"""CPU oracle for the LAFF retrieval hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy (fp32) restatement of the reference's arithmetic, written from the
formulas, one function per reference symbol.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
import this module, and only as the checker / the timed CPU baseline.  The
product path (`laff_amd/`) never imports it and has no CPU fallback.

Parity pinning: the reference ships no golden vectors or tests for this path
(SURVEY.md section 4), so the oracle is pinned against fixtures produced by running the
reference's own Python in the build container (`tools/gen_golden.py` ->
`tests/golden/*.npz`); `tests/test_oracle_golden.py` checks every function here
against them (max |diff| <= 2e-6 on unit-norm outputs, metrics exact).

All citations are file:line under /root/reference.
"""
import numpy as np

F32 = np.float32


def _f32(a):
    return np.ascontiguousarray(a, dtype=F32)


# ------------------------------------------------------------------------------------------------
# loss.py / evaluation.py primitives
# ------------------------------------------------------------------------------------------------
def l2norm(X, eps=1e-13, axis=1):
    """loss.l2norm (loss.py:8-13): X / (sqrt(sum X^2) + eps + 1e-14), fp32."""
    X = _f32(X)
    norm = np.sqrt(np.sum(X * X, axis=axis, keepdims=True, dtype=F32)).astype(F32)
    norm = norm + F32(eps) + F32(1e-14)
    return (X / norm).astype(F32)


def cosine_sim(query, retrio):
    """loss.cosine_sim (loss.py:30-34): re-normalise both operands, then mm."""
    return (l2norm(query) @ l2norm(retrio).T).astype(F32)


def np_l2norm(X):
    """evaluation.l2norm (evaluation.py:11-16): X / (||X|| + 1e-10)."""
    norm = np.linalg.norm(X, axis=1, keepdims=True)
    return 1.0 * X / (norm + 1e-10)


def np_cosine_sim(q, r):
    """evaluation.cosine_sim (evaluation.py:44-49)."""
    return np_l2norm(q).dot(np_l2norm(r).T)


# ------------------------------------------------------------------------------------------------
# a1 / a2: TransformNet (model/model.py:211-276) and the no-transform branch (:1801-1805, :1822-1823)
# ------------------------------------------------------------------------------------------------
def batch_norm_eval(y, bn, eps=1e-5):
    """nn.BatchNorm1d in eval mode: (y - rm) / sqrt(rv + eps) * gamma + beta."""
    gamma, beta, rm, rv = (_f32(t) for t in bn)
    return ((y - rm) / np.sqrt(rv + F32(eps)) * gamma + beta).astype(F32)


def activation(y, name):
    if name in (None, False, '', 'none'):
        return y
    if name == 'tanh':
        return np.tanh(y).astype(F32)
    if name == 'relu':
        return np.maximum(y, F32(0)).astype(F32)
    if name == 'sigmoid':
        return (F32(1) / (F32(1) + np.exp(-y))).astype(F32)
    raise ValueError(name)


def transform_net(x, W=None, b=None, act='tanh', bn=None, tile_heads=1):
    """TransformNet.forward (model/model.py:257-276) in eval mode (dropout = identity).

    fc -> activation -> BN.  `tile_heads` > 1 restates the caller-side `x.repeat(1, heads)`
    of the no-transform branch (model/model.py:1822-1823, :1675-1676).
    """
    y = _f32(x)
    if tile_heads > 1:
        y = np.tile(y, (1, tile_heads))
    if W is not None:
        y = (y @ _f32(W).T).astype(F32)
        if b is not None:
            y = y + _f32(b)
    y = activation(y, act)
    if bn is not None:
        y = batch_norm_eval(y, bn)
    return _f32(y)


# ------------------------------------------------------------------------------------------------
# a6: Attention_1 (model/Attention.py:78-105)
# ------------------------------------------------------------------------------------------------
def attention_1(x, w, b, with_ave=False, mul=False, gw=1.0, return_weights=False):
    """x (N, L, d) -> (N, d): softmax_L(c.w + b) weighted sum (+ gw * mean) then l2norm(eps=0)."""
    x = _f32(x)
    w = _f32(w).reshape(-1)
    mean = x.mean(axis=1, dtype=F32)                         # raw_global_emb (:81)
    c = x * mean[:, None, :] if mul else x                   # (:83-86)
    logits = (c @ w).astype(F32) + F32(b)                    # (:88)
    logits = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(logits).astype(F32)
    a = (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)   # (:89)
    g = (a[:, :, None] * x)                                  # (:93)
    if with_ave:
        g = g + F32(gw) * mean[:, None, :]                   # (:94-99)
    g = g.sum(axis=1, dtype=F32)                             # (:101)
    out = l2norm(g, eps=0.0)                                 # (:103)
    if return_weights:
        return out, a
    return out


def just_average(x):
    """JustAverage.forward (model/Attention.py:35-37)."""
    return _f32(x).mean(axis=1, dtype=F32)


# ------------------------------------------------------------------------------------------------
# a5: Multi_head_MyApply_Attention (model/Attention.py:508-531)
# ------------------------------------------------------------------------------------------------
def multi_head_attention(x, w, b, gw, H, with_ave=False, mul=False, split_head=True, l2norm_each_head=False):
    """x (N, L, D) -> (N, H, d).  w (H, d), b (H,), gw (H,)."""
    x = _f32(x)
    N, L, D = x.shape
    if split_head:
        d = D // H
        xh = x.reshape(N, L, H, d)                          # (:517)
    else:
        d = D
        xh = np.repeat(x[:, :, None, :], H, axis=2)        # (:520)
    if l2norm_each_head:
        xh = l2norm(xh, axis=3)                             # (:522-523), default eps
    outs = []
    w = _f32(w).reshape(H, d)
    for h in range(H):                                      # (:526-527)
        outs.append(attention_1(xh[:, :, h, :], w[h], np.asarray(b).reshape(-1)[h], with_ave, mul,
                                np.asarray(gw).reshape(-1)[h]))
    return np.stack(outs, axis=1)                           # (:529)


# ------------------------------------------------------------------------------------------------
# towers: a3/a4 (model/model.py:1807-1876, :1663-1705) and a7 (:2147-2190)
# ------------------------------------------------------------------------------------------------
def fuse_tower(feature_specs, att, H, expert=None, expert_l2norm=False):
    """feature_specs: list of dicts for transform_net (x, W, b, act, bn, tile_heads), in stack order.
    att: dict(w, b, gw, with_ave, mul, split_head, l2norm_each_head) or {'kind': 'just_average'}.
    expert: (L, D) expert-embedding rows added to the stacked planes (:1866-1870 / :1686-1690); expert_l2norm: then
    l2norm over D (:1872-1873 / :1693-1694)."""
    planes = [transform_net(**s) for s in feature_specs]
    local = np.stack(planes, axis=1)                        # torch.stack(dim=1) (:1862 / :1683)
    if expert is not None:
        local = (local + _f32(expert)[None, :local.shape[1], :]).astype(F32)
    if expert_l2norm:
        local = l2norm(local, axis=2)
    if att.get('kind') == 'just_average':
        return just_average(local)
    if att.get('kind') == 'attention_1':
        return attention_1(local, att['w'], att['b'], att['with_ave'], att['mul'], att['gw'])
    return multi_head_attention(local, att['w'], att['b'], att['gw'], H, att['with_ave'], att['mul'],
                                att.get('split_head', True), att.get('l2norm_each_head', False))


def frame_attention(frames, w, b, with_ave=False, mul=False, gw=1.0, Wfc=None, bfc=None):
    """Per-video Attention_1 over the F_max (zero padded) frames of the batch, exactly as the
    reference runs it (model/model.py:2167-2173): the `[0:n]` slice there acts on the size-1 batch
    axis, so every padded frame takes part; with `vis_frame_addFC` the Linear(512,512) is applied
    to the padded zeros as well (:2134-2138).  frames (B, Fmax, 512) -> (B, 512)."""
    frames = _f32(frames)
    B = frames.shape[0]
    out = np.empty((B, frames.shape[2]), F32)
    for i in range(B):
        x = frames[i:i + 1]
        if Wfc is not None:
            x = (x @ _f32(Wfc).T + _f32(bfc)).astype(F32)
        out[i] = attention_1(x, w, b, with_ave, mul, gw)[0]
    return out


# ------------------------------------------------------------------------------------------------
# a9-a11: similarity
# ------------------------------------------------------------------------------------------------
def txt2vis_matrix(txt_embs, vis_embs):
    """W2VVPP.get_txt2vis_matrix (model/model.py:1003-1016): 2-D -> cosine_sim; 3-D -> mean over heads."""
    t, v = _f32(txt_embs), _f32(vis_embs)
    if t.ndim == 2:
        return cosine_sim(t, v)
    sims = [cosine_sim(t[:, h, :], v[:, h, :]) for h in range(v.shape[1])]
    return np.mean(np.stack(sims, axis=0), axis=0, dtype=F32).astype(F32)


def txt2vis_matrix_fast(txt_embs, vis_embs):
    """Same value (to fp32 rounding) as one GEMM over the concatenated heads / H;
    valid when rows are already unit-norm per head, which the towers guarantee."""
    t, v = _f32(txt_embs), _f32(vis_embs)
    if t.ndim == 2:
        return (t @ v.T).astype(F32)
    H = t.shape[1]
    return ((t.reshape(t.shape[0], -1) @ v.reshape(v.shape[0], -1).T) / F32(H)).astype(F32)


def predict_blocked(txt_emb_batches, vis_emb_batches, Nt, Nv):
    """Loop B/C of W2VVPP.predict (model/model.py:1057-1077) on pre-computed embedding batches:
    (idxs_rows, emb) pairs; per (text block, video block) get_txt2vis_matrix, scatter into scores."""
    scores = np.zeros((Nt, Nv), F32)
    for rows, te in txt_emb_batches:
        for cols, ve in vis_emb_batches:
            scores[np.ix_(rows, cols)] = txt2vis_matrix(te, ve)
    return scores


def txt2vis_matrix_f64(txt_embs, vis_embs):
    """The INFINITELY PRECISE value of get_txt2vis_matrix on the given fp32 embeddings (model/model.py:1003-1016 on
    loss.py:30-34): mean_h <t_h, v_h> / ((|t_h| + 1e-13 + 1e-14)(|v_h| + 1e-13 + 1e-14)) evaluated in float64 (products of fp32
    values are exact in fp64).  The reference evaluates the same expression in fp32, i.e. this value +- ~1e-7; ranks taken on it are
    the tie-free limit of the reference's ranks.  Used as the rank oracle for the exact-rank pipeline."""
    t, v = np.asarray(txt_embs, dtype=np.float64), np.asarray(vis_embs, dtype=np.float64)
    if t.ndim == 2:
        t, v = t[:, None, :], v[:, None, :]
    H = t.shape[1]
    S = np.zeros((t.shape[0], v.shape[0]), np.float64)
    for h in range(H):
        tn = t[:, h, :] / (np.sqrt((t[:, h, :] ** 2).sum(axis=1, keepdims=True)) + (1e-13 + 1e-14))
        vn = v[:, h, :] / (np.sqrt((v[:, h, :] ** 2).sum(axis=1, keepdims=True)) + (1e-13 + 1e-14))
        S += tn @ vn.T
    return S / H


def count_ranks(S, gt_cols):
    """rank[t] = 1 + #{v != gt : S[t, v] > S[t, gt]} (predictor.py:232-244 in count form, single ground truth per row)."""
    S = np.asarray(S)
    gt_cols = np.asarray(gt_cols).astype(np.int64)
    sg = S[np.arange(S.shape[0]), gt_cols]
    above = S > sg[:, None]
    above[np.arange(S.shape[0]), gt_cols] = False
    return above.sum(axis=1).astype(np.int64) + 1


# ------------------------------------------------------------------------------------------------
# a12-a14: ranks and metrics
# ------------------------------------------------------------------------------------------------
def gt_positions(score_row, gt_cols):
    """1-based positions of the GT columns in the descending order of one score row -- what
    `np.where(label_matrix[i] == 1)[0] + 1` yields after predictor.py:239-243.  Count-based so that it does not
    depend on argsort's tie order: pos = 1 + #{c != g : s_c > s_g} (+ GT's own rank among GTs)."""
    s = np.asarray(score_row)
    gt_cols = np.asarray(gt_cols)
    sg = np.sort(s[gt_cols])[::-1]
    pos = []
    for i, val in enumerate(sg):
        pos.append(int(np.sum(s > val)) + 1 + (i - int(np.sum(sg[:i] > val))))
    return np.array(sorted(pos))


def eval_from_positions(positions):
    """evaluation.eval (evaluation.py:92-109) given per-row sorted 1-based GT positions."""
    ranks = np.array([p[0] for p in positions], dtype=np.float64)
    aps = np.array([np.mean([(i + 1.) / p[i] for i in range(len(p))]) for p in positions])
    r1, r5, r10 = [100.0 * np.mean(ranks <= k) for k in (1, 5, 10)]
    medr = np.floor(np.median(ranks))
    return (r1, r5, r10, medr, ranks.mean(), (1.0 / ranks).mean(), aps.mean())


def eval_label_matrix(label_matrix):
    """evaluation.eval (evaluation.py:92-109) on a 0/1 label matrix."""
    lab = np.asarray(label_matrix).astype(int)
    return eval_from_positions([np.where(row == 1)[0] + 1 for row in lab])


def predictor_metrics(scores, txt_ids, vis_ids):
    """predictor.py:232-246 (T2V) and :262-270 (V2T): id matching on `txt_id.split('#')[0]`."""
    vis_index = {v: i for i, v in enumerate(vis_ids)}
    owner = np.array([vis_index[t.split('#')[0]] for t in txt_ids])
    t2v = eval_from_positions([gt_positions(scores[i], [owner[i]]) for i in range(scores.shape[0])])
    v2t_pos = []
    for v in range(scores.shape[1]):
        v2t_pos.append(gt_positions(scores[:, v], np.where(owner == v)[0]))
    v2t = eval_from_positions(v2t_pos)
    return t2v, v2t


def eval_qry2retro(sim, n_qry=1):
    """evaluation.eval_qry2retro (evaluation.py:64-89): 0-based ranks, medr = floor(median)+1."""
    sim = np.asarray(sim)
    assert sim.shape[0] / sim.shape[1] == n_qry
    ranks = np.zeros(sim.shape[0])
    for i in range(sim.shape[0]):
        g = int(i / n_qry) if n_qry == 1 else None
        if g is None or i / n_qry != g:
            raise ValueError('only meaningful for n_qry == 1 (true division in the reference)')
        ranks[i] = np.sum(sim[i] > sim[i, g])
    r1 = 100.0 * np.sum(ranks < 1) / len(ranks)
    r5 = 100.0 * np.sum(ranks < 5) / len(ranks)
    r10 = 100.0 * np.sum(ranks < 10) / len(ranks)
    return (r1, r5, r10, np.floor(np.median(ranks)) + 1, ranks.mean() + 1, (1.0 / (ranks + 1)).mean())


# ------------------------------------------------------------------------------------------------
# state_dict -> specs (key names of SURVEY Appendix C)
# ------------------------------------------------------------------------------------------------
def _bn_from_sd(sd, prefix):
    if prefix + 'bn1.weight' not in sd:
        return None
    return (sd[prefix + 'bn1.weight'], sd[prefix + 'bn1.bias'], sd[prefix + 'bn1.running_mean'],
            sd[prefix + 'bn1.running_var'])


def feature_spec(sd, prefix, x, act, H, no_transform):
    bn = _bn_from_sd(sd, prefix)
    if no_transform:
        return dict(x=x, W=None, b=None, act=None, bn=bn, tile_heads=H)
    return dict(x=x, W=sd[prefix + 'fc1.weight'], b=sd[prefix + 'fc1.bias'], act=act, bn=bn, tile_heads=1)


def attention_from_sd(sd, prefix, H, with_ave, mul, split_head=True, l2norm_each_head=False):
    w = np.stack([sd['%sattention_layer.%d.embedding_common.0.weight' % (prefix, h)].reshape(-1) for h in range(H)])
    b = np.array([sd['%sattention_layer.%d.embedding_common.0.bias' % (prefix, h)].reshape(()) for h in range(H)], F32)
    gw = np.array([sd['%sattention_layer.%d.global_emb_weight_net.weight' % (prefix, h)].reshape(())
                   for h in range(H)], F32)
    return dict(w=w, b=b, gw=gw, with_ave=with_ave, mul=mul, split_head=split_head,
                l2norm_each_head=l2norm_each_head)


FRAME_ATTENTION_FLAGS = {  # model/model.py:94-97 (name -> with_ave, mul)
    'attention_noAverageMul_Ave': (True, False),
    'attention_noAveNoAverageMul': (False, False),
    'attention_averageMul': (True, True),
    'average_AverageMul_noAve': (False, True),
}



# ---- text-side feature producers ---------------------------------------------------------
def tokenize(text, clean=True, remove_stopword=False, stopwords=()):
    """textlib.TextTool.tokenize, English branch (textlib.py:27-45): CR -> space, every non [A-Za-z0-9] -> space,
    strip, lower, whitespace split; optional stop-word removal."""
    import re
    sent = text
    if clean:
        sent = sent.replace('\r', ' ')
        sent = re.sub(r"[^A-Za-z0-9]", " ", sent).strip().lower()
    tokens = sent.split()
    if remove_stopword:
        tokens = [t for t in tokens if t not in stopwords]
    return tokens


def bow_encoding(text, vocab, remove_stopword=False, stopwords=()):
    """txt2vec.BowVec._encoding / BowVecNSW (txt2vec.py:56-63, :135-142): count vector over the vocabulary, norm=0."""
    index = {w: i for i, w in enumerate(vocab)}
    vec = np.zeros(len(vocab))
    for w in tokenize(text, True, remove_stopword, stopwords):
        i = index.get(w, -1)
        if i >= 0:
            vec[i] += 1
    return vec


def w2v_encoding(text, words, table, remove_stopword=False, stopwords=()):
    """txt2vec.W2Vec._encoding / W2VecNSW (txt2vec.py:97-104, :145-149): BigFile.read dedups the tokens and drops the
    unknown ones (bigfile.py:187-213), then the vectors are averaged; zeros if none is known."""
    index = {w: i for i, w in enumerate(words)}
    rows = sorted({index[w] for w in tokenize(text, True, remove_stopword, stopwords) if w in index})
    if not rows:
        return np.zeros(table.shape[1])
    return np.array([table[r].tolist() for r in rows]).mean(axis=0)


# ---- training loss ------------------------------------------------------------------------
def margin_ranking_loss(s, im, margin=0.2, max_violation=True, cost_style='sum', direction='t2i'):
    """loss.MarginRankingLoss.forward (loss.py:99-135) with measure='cosine', summed over heads like
    model/model.py:2037-2039, plus the gradients autograd gives the reference (restated analytically).

    s (captions), im (videos): (B, d) or (B, H, d).  Returns (loss float32, d_s, d_im)."""
    s = _f32(s)
    im = _f32(im)
    squeeze = s.ndim == 2
    if squeeze:
        s, im = s[:, None, :], im[:, None, :]
    B, H, d = s.shape
    total = F32(0.0)
    d_s = np.zeros_like(s)
    d_im = np.zeros_like(im)
    eye = np.eye(B, dtype=bool)
    for h in range(H):
        xs, xi = s[:, h, :], im[:, h, :]
        rs = np.sqrt((xs * xs).sum(1, keepdims=True)).astype(F32)
        ri = np.sqrt((xi * xi).sum(1, keepdims=True)).astype(F32)
        ns = rs + F32(1e-13) + F32(1e-14)
        ni = ri + F32(1e-13) + F32(1e-14)
        hs, hi = (xs / ns).astype(F32), (xi / ni).astype(F32)
        scores = (hi @ hs.T).astype(F32)                               # self.sim(im, s): rows = videos (loss.py:102)
        diag = np.diag(scores).copy()
        dS = np.zeros((B, B), F32)
        loss_h = F32(0.0)
        if direction in ('i2t', 'bidir'):
            cost = np.maximum(F32(margin) + scores - diag[:, None], F32(0)).astype(F32)
            cost[eye] = 0
            if max_violation:
                j = cost.argmax(1)
                v = cost[np.arange(B), j]
                w = F32(1.0 / B) if cost_style == 'mean' else F32(1)
                loss_h += (v.sum() * w) if cost_style == 'sum' else v.mean()
                act = v > 0
                np.add.at(dS, (np.arange(B)[act], j[act]), w)
                np.add.at(dS, (np.arange(B)[act], np.arange(B)[act]), -w)
            else:
                w = F32(1.0 / (B * B)) if cost_style == 'mean' else F32(1)
                loss_h += cost.sum() if cost_style == 'sum' else cost.mean()
                act = cost > 0
                dS += act * w
                dS[np.arange(B), np.arange(B)] -= act.sum(1) * w
        if direction in ('t2i', 'bidir'):
            cost = np.maximum(F32(margin) + scores - diag[None, :], F32(0)).astype(F32)
            cost[eye] = 0
            if max_violation:
                i = cost.argmax(0)
                v = cost[i, np.arange(B)]
                w = F32(1.0 / B) if cost_style == 'mean' else F32(1)
                loss_h += (v.sum() * w) if cost_style == 'sum' else v.mean()
                act = v > 0
                np.add.at(dS, (i[act], np.arange(B)[act]), w)
                np.add.at(dS, (np.arange(B)[act], np.arange(B)[act]), -w)
            else:
                w = F32(1.0 / (B * B)) if cost_style == 'mean' else F32(1)
                loss_h += cost.sum() if cost_style == 'sum' else cost.mean()
                act = cost > 0
                dS += act * w
                dS[np.arange(B), np.arange(B)] -= act.sum(0) * w
        total = F32(total + F32(loss_h))
        g_hi = dS @ hs                                                   # d loss / d normalised videos
        g_hs = dS.T @ hi
        d_im[:, h, :] = g_hi / ni - hi * ((hi * g_hi).sum(1, keepdims=True) / ri)
        d_s[:, h, :] = g_hs / ns - hs * ((hs * g_hs).sum(1, keepdims=True) / rs)
    if squeeze:
        d_s, d_im = d_s[:, 0, :], d_im[:, 0, :]
    return F32(total), d_s.astype(F32), d_im.astype(F32)
