"""The K = 512 similarity strips with shares of the work in proportion to the speed of each XCD (LAFF_XCD_W) against equal shares, on
one box, interleaved.  The kernel writes the wall-clock ticks each persistent workgroup spent (LAFF_STRIP_TIMES_PTR) -- workgroup b runs
on XCD b % 8."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev = 'cuda'
Nt, Nv, K = 40000, 10000, 512
t = torch.nn.functional.normalize(torch.randn(Nt, K, device=dev), dim=1); v = torch.nn.functional.normalize(torch.randn(Nv, K, device=dev), dim=1)
T = ops.pack_rows(t, True, 1e-13, 'fp16'); V = ops.pack_rows(v, True, 1e-13, 'fp16')
S = torch.empty(Nt, Nv, device=dev)
times = torch.zeros(512, dtype=torch.int64, device=dev)
os.environ['LAFF_STRIP_TIMES_PTR'] = str(times.data_ptr())
fn = lambda: ops.sim_gemm(T, V, out=S)

def run(w, n=600, reads=6):
    os.environ['LAFF_XCD_W'] = ','.join('%.5f' % x for x in w)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    acc = torch.zeros(8, dtype=torch.float64)
    tot = 0.0
    for r in range(reads):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n // reads): fn()
        e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
        tt = times[:256].cpu().double().reshape(32, 8)          # [b // 8][b % 8]
        acc += tt.mean(0) * 0.01                               # us
    return tot / (n // reads * reads), acc / reads

w = [1.0] * 8
ms, tx = run(w)
print('equal shares: %.4f ms; mean workgroup time by XCD (us): %s' % (ms, ' '.join('%.1f' % x for x in tx)))
for it in range(3):
    m = float(tx.mean())
    w = [w[x] * (m / float(tx[x])) for x in range(8)]
    s = sum(w) / 8
    w = [x / s for x in w]
    ms, tx = run(w)
    print('iteration %d: weights %s' % (it, ' '.join('%.4f' % x for x in w)))
    print('              %.4f ms; by XCD (us): %s' % (ms, ' '.join('%.1f' % x for x in tx)))
print('interleaved:')
for r in range(4):
    a, _ = run([1.0] * 8)
    b, _ = run(w)
    print('   equal %.4f ms   weighted %.4f ms   (%.2f %%)' % (a, b, (b / a - 1) * 100))
