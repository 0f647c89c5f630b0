import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev='cuda'
Nt,Nv,K=40000,10000,512
t=torch.nn.functional.normalize(torch.randn(Nt,K,device=dev),dim=1); v=torch.nn.functional.normalize(torch.randn(Nv,K,device=dev),dim=1)
T=ops.pack_rows(t,True,1e-13,'fp16'); V=ops.pack_rows(v,True,1e-13,'fp16')
S=torch.empty(Nt,Nv,device=dev)
gt=(torch.arange(Nt,device=dev)%Nv).to(torch.int32); sg=ops.row_dot_gt(T,V,gt,1); cnt=torch.zeros(Nt,dtype=torch.int32,device=dev)
def timeit(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
for _ in range(2):
    print('S only                 %.4f ms'%timeit(lambda: ops.sim_gemm(T,V,out=S)))
    print('S + fused count        %.4f ms'%timeit(lambda: ops.sim_gemm(T,V,out=S,gt_col=gt,s_gt=sg,count=cnt)))
    print('count only (no S)      %.4f ms'%timeit(lambda: ops.sim_gemm(T,V,want_scores=False,gt_col=gt,s_gt=sg,count=cnt)))
