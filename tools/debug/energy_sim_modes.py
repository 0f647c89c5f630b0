#!/usr/bin/env python3
"""Joules per launch / pJ per pair of the K = 512 similarity at C4's shape in its three modes, stand-alone loops of 3 s each with the
package energy counter around them: plain (scores only, no rank count), banded + scores (the headline), banded count-only.
The band test's price = banded + S minus plain; the score matrix's = banded + S minus count-only."""
import os, re, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops, synth, retrieval
import laff_amd.model.model as M


def energy_uj():
    t = subprocess.run(['rocm-smi', '--showenergycounter'], capture_output=True, text=True, timeout=10).stdout
    m = re.search(r'Accumulated Energy \(uJ\): *([0-9.eE+]+)', t)
    return float(m.group(1)) if m else None


M.FC_PRECISION = 'fp16x3'
dev = torch.device('cuda:0')
Nt, Nv, H, d, _ = synth.WORKLOADS['c4_40kx10k']
model = synth.build_model(H, d, dev, seed=1237)
vis, txt, gt, _ = synth.make_features(Nt, Nv, dev, seed=1237)
with torch.no_grad():
    ve, te = retrieval.embed(model, vis, txt)
T = ops.pack_rows(te, True, 1e-13, 'fp16'); V = ops.pack_rows(ve, True, 1e-13, 'fp16')
S = ops.alloc_scores(Nt, Nv, dev)
st = ops.rank_prepare(te, ve, T, V, gt)


def banded(ws):
    st.pairs[:4].zero_(); st.count.zero_()
    ops.sim_gemm_banded(st, want_scores=ws, out=S if ws else None)


modes = [('plain + S (no rank count)', lambda: ops.sim_gemm(T, V, out=S)), ('banded + S (headline)', lambda: banded(True)),
         ('banded count-only', lambda: banded(False))]
res = {}
for name, fn in modes:
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    e0, t0, n = energy_uj(), time.perf_counter(), 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(100):
            fn()
        n += 100
        torch.cuda.synchronize()
    el, e1 = time.perf_counter() - t0, energy_uj()
    j = (e1 - e0) * 1e-6 / n
    res[name] = (1e3 * el / n, j)
    print('%-28s %.4f ms  %.4f J  %.0f W  %.3f pJ per pair' % (name, 1e3 * el / n, j, j / (el / n), 1e12 * j / (float(Nt) * Nv)))
a, b, c = res['plain + S (no rank count)'], res['banded + S (headline)'], res['banded count-only']
print('band test + dumps: %+.4f ms %+.4f J (%.3f pJ per pair);  score matrix: %+.4f ms %+.4f J (%.1f pJ per byte)' % (
    b[0] - a[0], b[1] - a[1], 1e12 * (b[1] - a[1]) / (float(Nt) * Nv), b[0] - c[0], b[1] - c[1], 1e12 * (b[1] - c[1]) / (4.0 * Nt * Nv)))
