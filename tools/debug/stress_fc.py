"""Stress of the grouped fused-split FC launch against the single-problem fp32 FC: every output element of every repetition is compared.
   python tools/debug/stress_fc.py [reps]        (library: LAFF_HIP_LIB or the in-tree build)"""
import sys, collections, torch
sys.path.insert(0, '.')
from laff_amd import ops
dev = 'cuda'
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)
total = 0
for name, rows, din, dout in (('C2', [3000] * 4 + [10000] * 4, 512, 512), ('C4-like', [20000] * 2 + [10000] * 3, 512, 2048),
                              ('ragged', [257, 1000, 4099, 513, 129, 7000], 768, 512)):
    b = torch.randn(dout, device=dev) * 0.1
    sc = torch.rand(dout, device=dev) + 0.5
    sh = torch.randn(dout, device=dev) * 0.1
    W = [torch.randn(dout, din, device=dev) / 22 for _ in rows]
    Ws = [ops.split_rows(w) for w in W]
    X = [torch.randn(n, din, device=dev) for n in rows]
    for act in ('tanh', 'relu', 'none', 'sigmoid'):
        probs = [dict(x=X[i], weight_split=Ws[i], bias=b, bn_scale=sc, bn_shift=sh, activation=act) for i in range(len(rows))]
        refs = [ops.fc_act_bn(X[i], W[i], b, sc, sh, act) for i in range(len(rows))]
        nbad, worst = 0, 0.0
        for rep in range(reps):
            outs = ops.fc_act_bn_split_grouped(probs)
            for o, r in zip(outs, refs):
                d = (o - r).abs()
                nbad += int((d > 1e-3).sum())
                worst = max(worst, d.max().item())
        total += nbad
        print('%-8s %-8s reps %d  bad elements %d  max |diff| %.2e' % (name, act, reps, nbad, worst), flush=True)
print('TOTAL bad', total)
sys.exit(1 if total else 0)
