"""Strip kernel vs tiled kernel below / around the dispatch threshold (banded, with and without S).  LAFF_STRIP=2 forces the strip form
down to 8 x CUs units; LAFF_STRIP=0 never uses it."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, '.')
    import torch
    from laff_amd import ops
    dev = torch.device('cuda:0')
    for Nt, Nv in [(10000, 3000), (10000, 2990), (20000, 4000), (16000, 6000), (40000, 2000), (59800, 2990)]:
        g = torch.Generator(device=dev); g.manual_seed(3)
        z = torch.randn(Nv, 48, generator=g, device=dev); P1 = torch.randn(48, 512, generator=g, device=dev)
        gt = (torch.arange(Nt, device=dev) % Nv).to(torch.int32)
        Ev = (z @ P1 + 9 * torch.randn(Nv, 512, generator=g, device=dev)).reshape(Nv, 1, 512).contiguous()
        Et = (z[gt.long()] @ P1 + 9 * torch.randn(Nt, 512, generator=g, device=dev)).reshape(Nt, 1, 512).contiguous()
        T, V = ops.pack_rows(Et, True, 1e-13, 'fp16'), ops.pack_rows(Ev, True, 1e-13, 'fp16')
        st = ops.rank_prepare(Et, Ev, T, V, gt)
        S = ops.alloc_scores(Nt, Nv, dev)
        def timeit(fn, n=40):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e3
        def f1():
            st.pairs[:4].zero_(); ops.sim_gemm_banded(st, out=S)
        def f2():
            st.pairs[:4].zero_(); ops.sim_gemm_banded(st, want_scores=False)
        print('LAFF_STRIP=%s  %6d x %5d  units/CU %5.1f   with S %7.1f us   count-only %7.1f us' % (
            os.environ.get('LAFF_STRIP'), Nt, Nv, ((Nt + 255) // 256) * ((Nv + 31) // 32) / 256.0, timeit(f1), timeit(f2)), flush=True)
else:
    for m in ('0', '2'):
        subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, LAFF_STRIP=m))
