import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev='cuda'
torch.manual_seed(0)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
D=512
rows=[40000]*4+[10000]*4
W=[torch.randn(D,512,device=dev)/22 for _ in rows]
Ws=[ops.split_rows(w) for w in W]
X=[torch.randn(n,512,device=dev) for n in rows]
Xs=ops.split_rows_grouped(X)
b=torch.randn(D,device=dev)*0.1; sc=torch.rand(D,device=dev)+0.5; sh=torch.randn(D,device=dev)*0.1
outs=[torch.empty(n,D,device=dev) for n in rows]
flop=sum(2.0*n*512*D*3 for n in rows)
def run(act, bn, sub=None, fused=False):
    idx=range(len(rows)) if sub is None else sub
    probs=[dict(x=X[i] if fused else Xs[i], weight_split=Ws[i], bias=b, bn_scale=sc if bn else None, bn_shift=sh if bn else None, activation=act, out=outs[i]) for i in idx]
    return lambda: ops.fc_act_bn_split_grouped(probs)
for name,fn,fl in [('8 problems tanh+bn',run('tanh',True),flop),('8 problems no act/bn',run(None,False),flop),
                ('4 x 40000 rows tanh+bn',run('tanh',True,range(4)),flop*0.8),('1 x 40000 rows',run('tanh',True,[0]),flop*0.2),
                ('4 x 10000 rows',run('tanh',True,range(4,8)),flop*0.2),
                ('FUSED 8 problems tanh+bn',run('tanh',True,None,True),flop),('FUSED 8 problems no act/bn',run(None,False,None,True),flop),
                ('FUSED 4 x 40000 tanh+bn',run('tanh',True,range(4),True),flop*0.8)]:
    ms=timeit(fn); print('%-26s %.4f ms  %.0f TF (x3 flops)'%(name,ms,fl/ms/1e9))
# the same contraction as ONE similarity-style GEMM: 200000 x 512 x (K=512 x3)
t=torch.nn.functional.normalize(torch.randn(200000,512,device=dev),dim=1); v=torch.nn.functional.normalize(torch.randn(512,512,device=dev),dim=1)
T=ops.pack_rows(t,True,1e-13,'fp16x3'); V=ops.pack_rows(v,True,1e-13,'fp16x3'); S=torch.empty(200000,512,device=dev)
ms=timeit(lambda: ops.sim_gemm(T,V,out=S)); print('sim_gemm fp16x3 200000x512x512  %.4f ms  %.0f TF'%(ms,2.0*200000*512*512*3/ms/1e9))
t2=torch.nn.functional.normalize(torch.randn(16384,512,device=dev),dim=1); v2=torch.nn.functional.normalize(torch.randn(16384,512,device=dev),dim=1)
T2=ops.pack_rows(t2,True,1e-13,'fp16x3'); V2=ops.pack_rows(v2,True,1e-13,'fp16x3'); S2=torch.empty(16384,16384,device=dev)
ms=timeit(lambda: ops.sim_gemm(T2,V2,out=S2)); print('sim_gemm fp16x3 16384x16384x512  %.4f ms  %.0f TF'%(ms,2.0*16384*16384*512*3/ms/1e9))
