"""Per-wave barrier waits in the K loop of the fused-split FC (debug library with LAFF_GEMM_TRACE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from laff_amd import ops
dev = 'cuda'
torch.manual_seed(0)
rows = [40000] * 4 + [10000] * 4
W = [torch.randn(512, 512, device=dev) / 22 for _ in rows]
Ws = [ops.split_rows(w) for w in W]
X = [torch.randn(n, 512, device=dev) for n in rows]
b = torch.randn(512, device=dev) * 0.1
probs = [dict(x=X[i], weight_split=Ws[i], bias=b, activation='tanh') for i in range(8)]
for _ in range(3): ops.fc_act_bn_fused_grouped(probs)
torch.cuda.synchronize()
nb = sum(((n + 255) // 256) * 2 for n in rows)
tr = torch.zeros(nb * 32, dtype=torch.int64, device=dev)
os.environ['LAFF_GEMM_TRACE_PTR'] = str(tr.data_ptr())
ops.fc_act_bn_fused_grouped(probs); torch.cuda.synchronize()
full = tr.cpu().numpy()
w = full[nb * 8:nb * 32].reshape(nb, 24).astype(float)
print('barrier wait per K-step by wave (mean over tiles):', np.round(w[:, :8].mean(0) / 15))
print('vmcnt+lgkmcnt wait per K-step by wave          :', np.round(w[:, 8:16].mean(0) / 15))
arr = w[:, 16:24]; arr = arr - arr.min(1, keepdims=True)
print('arrival at the barrier of K-step 7 relative to the first wave:', np.round(arr.mean(0)))
