import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev = 'cuda'
torch.manual_seed(0)
N, H, d, Dk, k = 59800, 8, 512, 7811, 14
mode = sys.argv[1] if len(sys.argv) > 1 else 'gather'
wt = torch.randn(Dk, H * d, device=dev) / 30
bias = torch.randn(H * d, device=dev) * 0.1
sc = torch.rand(H * d, device=dev) + 0.5; sh = torch.randn(H * d, device=dev) * 0.1
clip = torch.randn(N, d, device=dev)
w = torch.randn(H, d, device=dev) * 0.2; b = torch.randn(H, device=dev) * 0.1; gw = torch.full((H,), 0.6, device=dev)
flags = ops.attention_flags(with_ave=True)
crow = (torch.arange(N + 1, device=dev) * k).int()
col = torch.clamp((Dk ** torch.rand(N * k, device=dev) - 1.0).long(), 0, Dk - 1).int()
csr = torch.sparse_csr_tensor(crow, col, torch.ones(N * k, device=dev), size=(N, Dk))
y = torch.randn(N, H * d, device=dev)
for _ in range(3):
    if mode == 'gather':
        ops.fuse([(None, False, sc, sh, 'tanh', (csr, wt, bias)), (clip, True, sc, sh)], H, d, w, b, gw, flags, packed_precision='fp16')
    elif mode == 'dense':
        ops.fuse([(y, False, None, None), (clip, True, sc, sh)], H, d, w, b, gw, flags, packed_precision='fp16')
    else:
        ops.fuse([(clip, True, sc, sh), (clip, True, sc, sh)], H, d, w, b, gw, flags, packed_precision='fp16')
torch.cuda.synchronize()
