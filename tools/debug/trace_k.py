import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from laff_amd import ops
dev='cuda'
Nt,Nv,K=[int(x) for x in sys.argv[1:4]]
torch.manual_seed(0)
t=torch.nn.functional.normalize(torch.randn(Nt,K,device=dev),dim=1); v=torch.nn.functional.normalize(torch.randn(Nv,K,device=dev),dim=1)
T=ops.pack_rows(t,True,1e-13,'fp16'); V=ops.pack_rows(v,True,1e-13,'fp16')
S=torch.empty(Nt,Nv,device=dev)
def run(): ops.sim_gemm(T,V,out=S)
for _ in range(3): run()
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/10
print('shape',Nt,Nv,K,'ms %.4f  TF %.1f'%(ms, 2.0*Nt*Nv*K/ms/1e9))
tile=256
nb=((Nt+tile-1)//tile)*((Nv+tile-1)//tile)
tr=torch.zeros(nb*16,dtype=torch.int64,device=dev)
os.environ['LAFF_GEMM_TRACE_PTR']=str(tr.data_ptr())
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms1=e0.elapsed_time(e1)
os.environ.pop('LAFF_GEMM_TRACE_PTR')
raw=tr.cpu().numpy().astype(np.int64); a=raw[:nb*8].reshape(nb,8); w=raw[nb*8:].reshape(nb,8)
print('s_memtime ticks per s_memrealtime tick (100 MHz): %.3f -> s_memtime rate %.1f MHz'%(((w[:,7]-w[:,6])/np.maximum(1,(w[:,5]-w[:,4]))).mean(), 100*((w[:,7]-w[:,6])/np.maximum(1,(w[:,5]-w[:,4]))).mean()))
print('wave0 waits per K-step: lds %.0f  vm(DMA landing) %.0f  barrier %.0f   (nkt %d; includes ~3 s_memtime round trips)'%(w[:,0].mean()/np.maximum(1,w[:,3]-1).mean(), w[:,1].mean()/np.maximum(1,w[:,3]-1).mean(), w[:,2].mean()/np.maximum(1,w[:,3]-1).mean(), w[0,3]))
span=a[:,6].max()-a[a[:,0]>0,0].min()
print('traced launch ms %.4f, span ticks %d -> counter rate %.1f MHz'%(ms1,span,span/ms1/1e3))
d=np.diff(a[:,:7],axis=1)
names=['setup','issue+prologue wait','kstep0','ksteps 1..','barrier','epilogue']
nk=K*2//128
for i,n in enumerate(names):
    print('%-22s mean %8.0f  p50 %8.0f'%(n,d[:,i].mean(),np.median(d[:,i])))
print('per k-step (steps 1..): %.0f ticks'%(d[:,3].mean()/max(1,nk-1)))
tot=a[:,6]-a[:,0]
print('WG total mean %.0f; sum/span = %.1f concurrent WGs'%(tot.mean(), tot.sum()/span))
