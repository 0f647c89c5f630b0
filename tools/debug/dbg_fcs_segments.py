"""fc_strip against the fp32 FC on shapes that produce short segments; prints where they differ."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from laff_amd import ops
dev = torch.device('cuda:0')
shapes = [(40000, 64), (40000, 96), (20000, 128), (33000, 160), (9000, 32), (70000, 32), (12800, 1024), (40000, 512)]
if len(sys.argv) > 2: shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for N, D in shapes:
    g = torch.Generator(device=dev); g.manual_seed(N + D)
    x = torch.randn(N, 512, device=dev, generator=g)
    W = torch.randn(D, 512, device=dev, generator=g) / 512 ** 0.5
    b = torch.randn(D, device=dev, generator=g) * 0.1
    sw = ops.fc_strip_pack(W, b, None, None, 'tanh')
    y = ops.fc_act_bn_strip_grouped([dict(x=x, strip=sw)])[0]
    y32 = ops.fc_act_bn(x, W, b, None, None, 'tanh')
    torch.cuda.synchronize()
    err = (y - y32).abs()
    bad = err > 2e-5
    print('N=%d D=%d max err %.3g bad %d nan %d' % (N, D, float(err.max()), int(bad.sum()), int(torch.isnan(y).sum())))
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        strips = torch.unique(rows // 128)
        print('   bad rows', rows.numel(), 'strips', strips[:20].tolist(), '... of', (N + 127) // 128)
        print('   bad rows mod 128 (first strip):', (rows[rows // 128 == strips[0]] % 128)[:40].tolist())
        cols = bad.any(0).nonzero().flatten()
        print('   bad cols', cols.numel(), cols[:40].tolist())
        r = int(rows[0]); c = int(bad[r].nonzero()[0])
        print('   e.g. y[%d,%d] = %.6f want %.6f' % (r, c, float(y[r, c]), float(y32[r, c])))
