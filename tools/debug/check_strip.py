"""The strip form of the K = 512 similarity GEMM (sim_strip.hip) against the tiled kernel and against float64 ranks.
   python tools/debug/check_strip.py [Nt Nv]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from laff_amd import ops  # noqa: E402


def set_mode(mode):
    os.environ['LAFF_STRIP'] = str(mode)
    ops._ctx.clear()           # LAFF_STRIP is read when a ctx is created


def run(Et, Ev, gt, prec, want_scores=True, reps=0):
    T, V = ops.pack_rows(Et, True, 1e-13, prec), ops.pack_rows(Ev, True, 1e-13, prec)
    S, count, st = ops.exact_ranks(Et, Ev, T, V, gt, want_scores)
    torch.cuda.synchronize()
    ms = None
    if reps:
        st2 = ops.rank_prepare(Et, Ev, T, V, gt)
        out = torch.empty_like(S) if S is not None else None
        for _ in range(3):
            ops.sim_gemm_banded(st2, want_scores, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.sim_gemm_banded(st2, want_scores, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
    return S, count, st, ms


def fp64_ranks(Et, Ev, gt):
    t, v = Et.double(), Ev.double()
    t = t / (t.pow(2).sum(-1, keepdim=True).sqrt() + 1.1e-13)
    v = v / (v.pow(2).sum(-1, keepdim=True).sqrt() + 1.1e-13)
    out = torch.empty(Et.shape[0], dtype=torch.int32, device=Et.device)
    for a in range(0, Et.shape[0], 4096):
        S = torch.einsum('thd,vhd->tv', t[a:a + 4096], v) / Et.shape[1]
        g = gt[a:a + 4096].long()
        ab = S > S.gather(1, g[:, None])
        ab[torch.arange(ab.shape[0], device=S.device), g] = False
        out[a:a + 4096] = ab.sum(1).to(torch.int32)
    return out


def main():
    Nt, Nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40000, 10000)
    reps = int(os.environ.get('REPS', '20'))
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(3)
    noise = float(os.environ.get('NOISE', '9'))
    z = torch.randn(Nv, 48, generator=g, device=dev)
    P1 = torch.randn(48, 512, generator=g, device=dev)
    gt = (torch.arange(Nt, device=dev) % Nv).to(torch.int32)
    Ev = (z @ P1 + noise * torch.randn(Nv, 512, generator=g, device=dev)).reshape(Nv, 1, 512).contiguous()
    Et = (z[gt.long()] @ P1 + noise * torch.randn(Nt, 512, generator=g, device=dev)).reshape(Nt, 1, 512).contiguous()
    want = fp64_ranks(Et, Ev, gt)
    print('ranks: R@1 %.2f  mean %.1f  max %d' % (100.0 * (want == 0).float().mean(), want.float().mean(), int(want.max())))
    for prec in ('fp16', 'bf16'):
        set_mode(0)
        S0, c0, st0, ms0 = run(Et, Ev, gt, prec, True, reps)
        _, _, _, ms0n = run(Et, Ev, gt, prec, False, reps)
        p0 = st0.pair_indices()
        set_mode(1)
        for ws in (True, False):
            S1, c1, st1, ms1 = run(Et, Ev, gt, prec, ws, reps)
            ok_r = torch.equal(c1, want)
            line = '%s strip scores %d: ranks exact %s  flag %s  %.4f ms (tiled %.4f)' % (
                prec, ws, ok_r, int(st1.pairs[1]), ms1 or 0, (ms0 if ws else ms0n) or 0)
            if ws:
                d = (S1 - S0).abs().max().item()
                p1 = st1.pair_indices()
                k0 = set(map(tuple, p0.tolist())) if p0.shape[0] < 2000000 else None
                k1 = set(map(tuple, p1.tolist())) if p1.shape[0] < 2000000 else None
                line += '  max|S - S_tiled| %.3g  equal %s  pairs %d (tiled %d) same set %s' % (
                    d, torch.equal(S1, S0), p1.shape[0], p0.shape[0], k0 == k1 if k0 is not None else '?')
                if not torch.equal(S1, S0):
                    bad = (S1 != S0).nonzero()
                    print('   differing entries', bad.shape[0], 'first', bad[:6].tolist())
            print(line, flush=True)
            if not ok_r:
                bad = (c1 != want).nonzero().flatten()
                print('   bad rows', bad.numel(), 'first', bad[:8].tolist(), 'got', c1[bad[:8]].tolist(), 'want', want[bad[:8]].tolist())
        print('%s tiled: ranks exact %s' % (prec, torch.equal(c0, want)))
        # scores only (no count)
        T, V = ops.pack_rows(Et, True, 1e-13, prec), ops.pack_rows(Ev, True, 1e-13, prec)
        set_mode(0)
        Sp0 = ops.sim_gemm(T, V)
        set_mode(1)
        Sp1 = ops.sim_gemm(T, V)
        print('%s plain scores: equal to tiled %s' % (prec, torch.equal(Sp0, Sp1)), flush=True)


if __name__ == '__main__':
    main()
