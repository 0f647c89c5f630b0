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
for N in (40000,) * 4 + (10000,) * 4:
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
t0 = a[:, 0].min()
ends = seg[:, :, 6].max(axis=1)
print('kernel span (cycles): %d   starts spread %d   end spread %d' % (ends.max() - t0, a[:, 0].max() - t0, ends.max() - ends.min()))
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
