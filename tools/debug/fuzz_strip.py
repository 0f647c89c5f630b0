"""Random shapes: the strip kernel (LAFF_STRIP=3) against the tiled kernel (0): scores bit-equal, counts equal, no overflow.
   python tools/debug/fuzz_strip.py [n_cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from laff_amd import ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda')


def mode(m):
    os.environ['LAFF_STRIP'] = str(m)
    ops.reset_contexts()


bad = 0
for case in range(n_cases):
    # at least 2048 (strip, block) units so that the strip kernel is eligible in mode 3
    while True:
        Nt = int(rng.integers(300, 30000))
        Nv = int(rng.integers(300, 30000))
        if ((Nt + 255) // 256) * ((Nv + 31) // 32) >= 2100 and Nt * Nv <= 3e8:
            break
    prec = 'fp16' if rng.random() < 0.7 else 'bf16'
    scores = rng.random() < 0.7
    ldo = Nv + int(rng.choice([0, 0, 4, 12, 16, 36]))
    ldo = (ldo + 3) & ~3
    if ldo < Nv:
        ldo += 4
    noise = float(rng.choice([0.5, 3.0, 9.0]))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    z = torch.randn(Nv, 24, generator=g, device=dev)
    P = torch.randn(24, 512, generator=g, device=dev)
    gt = torch.randint(0, Nv, (Nt,), generator=g, device=dev).to(torch.int32)
    Ev = (z @ P + noise * torch.randn(Nv, 512, generator=g, device=dev)).reshape(Nv, 1, 512).contiguous()
    Et = (z[gt.long()] @ P + noise * torch.randn(Nt, 512, generator=g, device=dev)).reshape(Nt, 1, 512).contiguous()
    out = {}
    for m in (0, 3):
        mode(m)
        T, V = ops.pack_rows(Et, True, 1e-13, prec), ops.pack_rows(Ev, True, 1e-13, prec)
        st = ops.rank_prepare(Et, Ev, T, V, gt)
        S = torch.full((Nt, ldo), -7.0, device=dev)[:, :Nv] if scores else None
        S = ops.sim_gemm_banded(st, scores, out=S)
        used = int(st._header()[2]) >> 31
        ops.rank_resolve(st, S)
        torch.cuda.synchronize()
        out[m] = (S, st.count.clone(), st.listed_pairs()[1], used, S._base if (S is not None and S._base is not None) else None)
    if out[3][2] or out[0][2]:
        # a list that was too small is reported (flag + poisoned count[0]), never silently wrong: nothing else to compare
        poisoned = all(int(o[1][0]) < -(1 << 25) for o in (out[0], out[3]) if o[2])
        print('%3d  %6d x %6d ldo %6d %s scores %d noise %.1f : pair list overflow (tiled %s, strip %s), poisoned %s' % (
            case, Nt, Nv, ldo, prec, scores, noise, out[0][2], out[3][2], poisoned), flush=True)
        bad += not poisoned
        continue
    ok = out[3][3] == 1 and torch.equal(out[0][1], out[3][1])
    if scores:
        ok = ok and torch.equal(out[0][0], out[3][0])
        if out[3][4] is not None and ldo > Nv:
            ok = ok and bool((out[3][4][:, Nv:] == -7.0).all())
    if not ok:
        print('     overflow tiled %s strip %s; counts equal %s; scores equal %s; max |count diff| %d' % (
            out[0][2], out[3][2], torch.equal(out[0][1], out[3][1]), (torch.equal(out[0][0], out[3][0]) if scores else None),
            int((out[0][1] - out[3][1]).abs().max())))
    bad += not ok
    print('%3d  %6d x %6d ldo %6d %s scores %d noise %.1f : %s' % (case, Nt, Nv, ldo, prec, scores, noise, 'ok' if ok else 'MISMATCH (strip ran: %d)' % out[3][3]), flush=True)
print('mismatches:', bad)
sys.exit(1 if bad else 0)
