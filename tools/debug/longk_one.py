import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev = "cuda"; N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384; K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
t = torch.nn.functional.normalize(torch.randn(N, K, device=dev), dim=1); v = torch.nn.functional.normalize(torch.randn(N, K, device=dev), dim=1)
T = ops.pack_rows(t, True, 1e-13, 'fp16'); V = ops.pack_rows(v, True, 1e-13, 'fp16')
S = torch.empty(N, N, device=dev)
a16, b16 = t.half(), v.half(); C = torch.empty(N, N, device=dev, dtype=torch.float16)
for _ in range(8): ops.sim_gemm(T, V, out=S)
torch.cuda.synchronize()
for _ in range(8): torch.matmul(a16, b16.t(), out=C)
torch.cuda.synchronize()
