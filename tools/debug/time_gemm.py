import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev='cuda'
for (Nt,Nv,K) in [(40000,10000,512),(16384,16384,4096)]:
    t=torch.nn.functional.normalize(torch.randn(Nt,K,device=dev),dim=1); v=torch.nn.functional.normalize(torch.randn(Nv,K,device=dev),dim=1)
    T=ops.pack_rows(t,True,1e-13,'fp16'); V=ops.pack_rows(v,True,1e-13,'fp16')
    S=torch.empty(Nt,Nv,device=dev)
    for _ in range(5): ops.sim_gemm(T,V,out=S)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.sim_gemm(T,V,out=S)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/50
    print(os.environ.get('LAFF_HIP_LIB','default'),Nt,Nv,K,'ms %.4f TF %.1f'%(ms,2.0*Nt*Nv*K/ms/1e9))
