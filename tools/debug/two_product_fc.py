#!/usr/bin/env python3
"""Gated experiment of the round-5 review, done in numpy (no kernel needed to see the outcome): the FC projection with TWO fp16 products,
X_hi (W_hi + W_lo), instead of the three of the hi/lo split, X_hi W_hi + X_hi W_lo + X_lo W_hi.  Operands are split exactly as
fc_strip.hip splits them (row / column maxima scaled into [512, 1024) by a power of two, hi = fp16(x), lo = fp16(x - hi)); products and
sums in float64, so what is measured is the operand error alone.  Towers: 4 features of 512-d per side through FC -> tanh, mean over the
features, L2-norm (the attention weights of the synthetic model are near-uniform); scores = cosine."""
import numpy as np

rng = np.random.default_rng(7)
Nt, Nv, K, D, L = 2000, 500, 512, 512, 4


def split(a, axis):
    m = np.abs(a).max(axis=axis, keepdims=True)
    s = 2.0 ** (9 - np.floor(np.log2(m)))          # maximum into [512, 1024)
    hi = (a * s).astype(np.float16).astype(np.float64)
    lo = ((a * s) - hi).astype(np.float16).astype(np.float64)
    return hi, lo, s


def tower(X, W, mode):
    outs = []
    for x, w in zip(X, W):
        xh, xl, sx = split(x, 1)
        wh, wl, sw = split(w, 1)
        if mode == 'exact':
            y = x @ w.T
        elif mode == 'x3':
            y = (xh @ wh.T + xh @ wl.T + xl @ wh.T) / (sx * sw.T)
        else:
            y = (xh @ (wh + wl).T) / (sx * sw.T)
        outs.append(np.tanh(y))
    g = sum(outs) / len(outs)
    return g / np.linalg.norm(g, axis=1, keepdims=True)


z = rng.normal(size=(Nv, 64))
gt = np.arange(Nt) % Nv
P = [rng.normal(size=(64, K)) for _ in range(L)]
Xv = [(z @ p + 0.5 * rng.normal(size=(Nv, K))).astype(np.float32).astype(np.float64) for p in P]
Xt = [(z[gt] @ p + 0.5 * rng.normal(size=(Nt, K))).astype(np.float32).astype(np.float64) for p in P]
lim = np.sqrt(6.0 / (K + D))
W = [rng.uniform(-lim, lim, size=(D, K)).astype(np.float32).astype(np.float64) for _ in range(L)]
ref_t, ref_v = tower(Xt, W, 'exact'), tower(Xv, W, 'exact')
S_ref = ref_t @ ref_v.T
for mode in ('x3', 'x2'):
    et, ev = tower(Xt, W, mode), tower(Xv, W, mode)
    S = et @ ev.T
    r_ref = (S_ref > S_ref[np.arange(Nt), gt][:, None]).sum(1)
    r = (S > S[np.arange(Nt), gt][:, None]).sum(1)
    print('%s: max |d emb| %.2e (tolerance of the tower tests 5e-6)   max |d cos| %.2e   ranks that differ from the exact path %d of %d'
          % ({'x3': 'three products (shipped)', 'x2': 'two products           '}[mode], max(np.abs(et - ref_t).max(), np.abs(ev - ref_v).max()),
             np.abs(S - S_ref).max(), int((r != r_ref).sum()), Nt))
