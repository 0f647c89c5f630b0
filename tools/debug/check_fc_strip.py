"""Strip-form FC (laff_fc_act_bn_strip_grouped) against fp64 and against the tiled fused-split path; timing at the C4 group.

    python tools/debug/check_fc_strip.py [--time]
"""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from laff_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def case(N, D, act, seed, with_bias=True, with_bn=True, ldy_pad=0, ldx_pad=0):
    g = np.random.default_rng(seed)
    x = g.normal(0, 1, (N, 512)).astype(np.float32)
    if N > 0:
        x[0] *= 3e4
    if N > 2:
        x[1] *= 1e-6
        x[2] = 0
    W = (g.normal(0, 1, (D, 512)) / np.sqrt(512)).astype(np.float32)
    b = g.normal(0, 0.1, D).astype(np.float32) if with_bias else None
    sc = g.uniform(0.5, 1.5, D).astype(np.float32) if with_bn else None
    sh = g.normal(0, 0.1, D).astype(np.float32) if with_bn else None
    xt = t(x)
    if ldx_pad:
        buf = torch.zeros((N, 512 + ldx_pad), device=dev)
        buf[:, :512] = xt
        xt = buf[:, :512]
    sw = ops.fc_strip_pack(t(W), t(b) if with_bias else None, t(sc) if with_bn else None, t(sh) if with_bn else None, act)
    out = None
    if ldy_pad:
        out = torch.full((N, D + ldy_pad), -7.0, device=dev)[:, :D]
    y = ops.fc_act_bn_strip_grouped([dict(x=xt, strip=sw, out=out)])[0]
    torch.cuda.synchronize()
    pre = x.astype(np.float64) @ W.astype(np.float64).T + (b if with_bias else 0.0)
    f = {None: lambda v: v, 'tanh': np.tanh, 'relu': lambda v: np.maximum(v, 0), 'sigmoid': lambda v: 1 / (1 + np.exp(-v))}[act]
    ref = f(pre) * (sc if with_bn else 1.0) + (sh if with_bn else 0.0)
    scale = 1.0 if act in ('tanh', 'sigmoid') else np.maximum(1.0, np.abs(pre).max(axis=1, keepdims=True))
    err = float(np.max(np.abs(y.cpu().numpy() - ref) / scale)) if N else 0.0
    ok = err <= 2e-5
    if ldy_pad and N:
        full = out._base if out._base is not None else out
        ok = ok and bool((full[:, D:] == -7.0).all())
    print('N=%6d D=%5d act=%-8s bias=%d bn=%d ldy+%d ldx+%d  max err %.2e  %s' % (N, D, act, with_bias, with_bn, ldy_pad, ldx_pad, err, 'ok' if ok else 'FAIL'))
    return ok


def grouped(seed=5):
    """the C4 group shape in small: 8 problems of two sizes in one launch, vs the tiled path"""
    g = np.random.default_rng(seed)
    probs, refs = [], []
    for i, N in enumerate((1000, 1000, 777, 130, 250, 250, 250, 31)):
        x = g.normal(0, 1, (N, 512)).astype(np.float32)
        W = (g.normal(0, 1, (512, 512)) / np.sqrt(512)).astype(np.float32)
        b = g.normal(0, 0.1, 512).astype(np.float32)
        sc = g.uniform(0.5, 1.5, 512).astype(np.float32)
        sh = g.normal(0, 0.1, 512).astype(np.float32)
        sw = ops.fc_strip_pack(t(W), t(b), t(sc), t(sh), 'tanh')
        probs.append(dict(x=t(x), strip=sw))
        refs.append(np.tanh(x.astype(np.float64) @ W.astype(np.float64).T + b) * sc + sh)
    ys = ops.fc_act_bn_strip_grouped(probs)
    torch.cuda.synchronize()
    err = max(float(np.abs(y.cpu().numpy() - r).max()) for y, r in zip(ys, refs))
    print('grouped 8 problems: max err %.2e %s' % (err, 'ok' if err <= 2e-5 else 'FAIL'))
    return err <= 2e-5


def timing():
    g = np.random.default_rng(1)
    probs_s, probs_f = [], []
    for N in (40000,) * 4 + (10000,) * 4:
        x = t(g.normal(0, 1, (N, 512)).astype(np.float32))
        W = t((g.normal(0, 1, (512, 512)) / np.sqrt(512)).astype(np.float32))
        b = t(g.normal(0, 0.1, 512).astype(np.float32))
        sc = t(g.uniform(0.5, 1.5, 512).astype(np.float32))
        sh = t(g.normal(0, 0.1, 512).astype(np.float32))
        out = torch.empty((N, 512), device=dev)
        probs_s.append(dict(x=x, strip=ops.fc_strip_pack(W, b, sc, sh, 'tanh'), out=out))
        probs_f.append(dict(x=x, weight_split=ops.split_rows(W), bias=b, bn_scale=sc, bn_shift=sh, activation='tanh', out=torch.empty_like(out)))
    for name, fn, pr in (('strip', ops.fc_act_bn_strip_grouped, probs_s), ('tiled fused (+row scales)', ops.fc_act_bn_fused_grouped, probs_f)):
        for _ in range(5):
            fn(pr)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn(pr)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        print('%-28s %.4f ms  (min of 5 x 20: %s)' % (name, min(ts), ' '.join('%.4f' % v for v in ts)))
    d = max(float((a['out'] - b['out']).abs().max()) for a, b in zip(probs_s, probs_f))
    print('strip vs tiled max diff %.2e' % d)


if __name__ == '__main__':
    ok = True
    for N, D, act in ((1, 32, 'tanh'), (31, 64, 'tanh'), (128, 512, 'tanh'), (129, 512, None), (300, 512, 'relu'), (1000, 512, 'sigmoid'),
                      (257, 4096, 'tanh'), (5000, 512, 'tanh'), (40000, 512, 'tanh')):
        ok &= case(N, D, act, N + D)
    ok &= case(333, 512, 'tanh', 3, with_bias=False, with_bn=False)
    ok &= case(333, 512, None, 4, with_bias=True, with_bn=False, ldy_pad=8)
    ok &= case(2000, 96, 'tanh', 6, ldy_pad=3, ldx_pad=4)
    ok &= grouped()
    print('ALL OK' if ok else 'FAILURES')
    if '--time' in sys.argv:
        timing()
    sys.exit(0 if ok else 1)
