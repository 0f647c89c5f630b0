import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from laff_amd import ops
dev='cuda'
g=np.random.default_rng(0)
for n,hi in [(40000,10000),(100000,30000),(59800,2990)]:
    r=g.integers(1,hi,n).astype(np.int32); r[:int(n*0.41)]=1
    t=torch.as_tensor(r,device=dev)
    out=torch.zeros(8,dtype=torch.float64).pin_memory()
    ops.ctx_prepare_metrics(t.device)
    for _ in range(3): ops.rank_metrics_async(t,out)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ops.rank_metrics_async(t,out)
    e1.record(); torch.cuda.synchronize()
    print(n,hi,'rank_metrics_async %.2f us per call (incl. 64-byte D2H)'%(e0.elapsed_time(e1)*10))
