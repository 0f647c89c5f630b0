"""Per-workgroup cycle stamps of the strip FC (debug build: tools/debug/build_fcs_variant.sh trace -DLAFF_FCS_TRACE, loaded through
LAFF_HIP_LIB=scratch/fcs_trace/liblaff_hip.so).   python tools/debug/trace_fc_strip.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from laff_amd import ops  # noqa: E402

dev = torch.device('cuda')
g = torch.Generator(device='cpu').manual_seed(1)
probs = []
PITCH = int(os.environ.get('FCS_PITCH', '512'))
NT_, NV_ = (int(v) for v in os.environ.get('FCS_SHAPE', '40000,10000').split(','))      # rows of the text / video features (4 each)
for N in (NT_,) * 4 + (NV_,) * 4:
    x = torch.randn((N, 512), generator=g).to(dev)
    if PITCH != 512:
        buf = torch.zeros((N, PITCH), device=dev)
        buf[:, :512] = x
        x = buf[:, :512]
    W = (torch.randn((512, 512), generator=g) / 512 ** 0.5).to(dev)
    b = (0.1 * torch.randn(512, generator=g)).to(dev)
    probs.append(dict(x=x, strip=ops.fc_strip_pack(W, b, None, None, 'tanh'), out=torch.empty((N, 512), device=dev)))
for _ in range(3):
    ops.fc_act_bn_strip_grouped(probs)
torch.cuda.synchronize()
nwg = 256
tr = torch.zeros(nwg * 80, dtype=torch.int64, device=dev)
os.environ['LAFF_GEMM_TRACE_PTR'] = str(tr.data_ptr())
ops.fc_act_bn_strip_grouped(probs)
torch.cuda.synchronize()
os.environ.pop('LAFF_GEMM_TRACE_PTR')
a = tr.cpu().numpy().reshape(nwg, 80).astype(np.int64)
seg = a[:, :64].reshape(nwg, 8, 8)
liv = a[:, 0] > 0
t0 = a[liv, 0].min()
ends = seg[:, :, 6].max(axis=1)
print('workgroups that stamped: %d; kernel span (cycles): %d   starts spread %d (p50 %d)   end spread %d' % (liv.sum(), ends[liv].max() - t0, a[liv, 0].max() - t0, np.median(a[liv, 0]) - t0, ends[liv].max() - ends[liv].min()))
names = ['strip -> registers (raw) + barrier', 'W prologue issue', 'converted', 'prologue landed + barrier', 'block loop', 'drain + segment end']
tot = np.zeros(6)
nseg = nblk = 0
for s in range(8):
    x = seg[:, s, :7]
    live = x[:, 6] > 0
    if not live.any():
        continue
    d = np.diff(x[live], axis=1)
    n = a[live, 64 + s]
    print('segment %d: %3d workgroups, blocks mean %5.1f | ' % (s, live.sum(), n.mean()) + '  '.join('%s %.0f' % (nm, v) for nm, v in zip(names, d.mean(axis=0))) +
          ' | loop per block %.0f' % (d[:, 4] / np.maximum(n, 1)).mean())
    tot += d.sum(axis=0)
    nseg += live.sum()
    nblk += n.sum()
print('totals per workgroup (cycles): ' + '  '.join('%s %.0f' % (nm, v / nwg) for nm, v in zip(names, tot)) + '  | sum %.0f' % (tot.sum() / nwg))
print('segments per workgroup %.2f, blocks per workgroup %.1f, loop cycles per block %.0f (MFMA issue alone: 3072)' % (nseg / nwg, nblk / nwg, tot[4] / nblk))
# per-workgroup busy time (first stamp of its first segment to last stamp of its last one: one clock domain per workgroup)
first = seg[:, 0, 0].astype(np.float64)
last = seg[:, :, 6].max(axis=1).astype(np.float64)
busy = (last - first)[liv]
xcd = np.arange(nwg)[liv] % 8
print('busy cycles per workgroup: mean %.0f  p50 %.0f  p95 %.0f  max %.0f  min %.0f' % (busy.mean(), np.median(busy), np.percentile(busy, 95), busy.max(), busy.min()))
print('by XCD (mean / max): ' + '  '.join('%d: %.0f / %.0f' % (x, busy[xcd == x].mean(), busy[xcd == x].max()) for x in range(8)))
