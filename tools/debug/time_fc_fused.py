import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laff_amd import ops
dev='cuda'
torch.manual_seed(0)
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best=[]
    for r in range(5):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1)/n)
    return sorted(best)[2], min(best)
D=512
rows=[40000]*4+[10000]*4
W=[torch.randn(D,512,device=dev)/22 for _ in rows]
Ws=[ops.split_rows(w) for w in W]
X=[torch.randn(n,512,device=dev) for n in rows]
b=torch.randn(D,device=dev)*0.1; sc=torch.rand(D,device=dev)+0.5; sh=torch.randn(D,device=dev)*0.1
outs=[torch.empty(n,D,device=dev) for n in rows]
probs=[dict(x=X[i], weight_split=Ws[i], bias=b, bn_scale=sc, bn_shift=sh, activation='tanh', out=outs[i]) for i in range(8)]
ms=timeit(lambda: ops.fc_act_bn_fused_grouped(probs))
print('LAFF_GEMM_VARIANT=%s fused grouped FC (scales + GEMM): median %.4f ms  min %.4f' % (os.environ.get('LAFF_GEMM_VARIANT','-'), ms[0], ms[1]))
