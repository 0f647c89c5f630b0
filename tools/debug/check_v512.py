import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev = 'cuda'
torch.manual_seed(0)
for (Nt, Nv, K) in ((4096, 4096, 512), (5000, 3001, 1024), (8192, 8192, 4096)):
    t = torch.nn.functional.normalize(torch.randn(Nt, K, device=dev), dim=1); v = torch.nn.functional.normalize(torch.randn(Nv, K, device=dev), dim=1)
    T = ops.pack_rows(t, True, 1e-13, 'fp16'); V = ops.pack_rows(v, True, 1e-13, 'fp16')
    S = ops.sim_gemm(T, V)
    ref = (t.half().double() @ v.half().double().t())
    print(Nt, Nv, K, 'max |S - fp64 product of the fp16 operands| = %.3e' % (S.double() - ref).abs().max().item())
