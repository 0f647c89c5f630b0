import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from laff_amd import ops
dev='cuda'
Nt,Nv,K=40000,10000,512
torch.manual_seed(0)
from laff_amd import synth, retrieval
_m = synth.build_model(1, 512, torch.device(dev), seed=1237)
_vis, _txt, gt, _ = synth.make_features(Nt, Nv, torch.device(dev), seed=1237)
with torch.no_grad():
    v, t = retrieval.embed(_m, _vis, _txt)          # the bench workload's own embeddings (C4): a realistic score distribution
t, v = t.reshape(Nt, K).contiguous(), v.reshape(Nv, K).contiguous()
prec=sys.argv[1] if len(sys.argv)>1 else 'fp16'
write=(len(sys.argv)<3 or 'count' not in sys.argv[2])
banded=len(sys.argv)>2 and 'banded' in sys.argv[2]
T=ops.pack_rows(t,True,1e-13,prec); V=ops.pack_rows(v,True,1e-13,prec)
S=torch.empty(Nt,Nv,device=dev)
sg=torch.ones(Nt,device=dev); cnt=torch.zeros(Nt,dtype=torch.int32,device=dev)
st=ops.rank_prepare(t,v,T,V,gt) if banded else None
def run():
    if banded: ops.sim_gemm_banded(st, want_scores=write, out=S if write else None)
    elif write: ops.sim_gemm(T,V,out=S)
    else: ops.sim_gemm(T,V,want_scores=False,gt_col=gt,s_gt=sg,count=cnt)
for _ in range(3): run()
torch.cuda.synchronize()
tile = 256 if os.environ.get('LAFF_GEMM_VARIANT','256')=='256' else 128
nb=((Nt+tile-1)//tile)*((Nv+tile-1)//tile)
tr=torch.zeros(nb*24,dtype=torch.int64,device=dev)
os.environ['LAFF_GEMM_TRACE_PTR']=str(tr.data_ptr())
run(); torch.cuda.synchronize()
os.environ.pop('LAFF_GEMM_TRACE_PTR')
full=tr.cpu().numpy().astype(np.int64)
a=full[:nb*8].reshape(nb,8)
e=full[nb*16:nb*24].reshape(nb,8)
if banded:
    de=np.diff(e,axis=1)
    for i,n in enumerate(['bcmax/setup','rowblk0','rowblk1','rowblk2','rowblk3','wait at barrier','flush']):
        print('  epi %-16s mean %8.0f p50 %8.0f p90 %8.0f'%(n,de[:,i].mean(),*np.percentile(de[:,i],[50,90])))
t0=a[:,0].min()
d=np.diff(a[:,:7],axis=1)
names=['setup','issue+prologue wait','kstep0','ksteps 1..','barrier','epilogue']
print('tile',tile,'nb',nb,'precision',prec,'write',write)
print('kernel span (ticks):', a[:,6].max()-t0)
for i,n in enumerate(names):
    print('%-22s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f'%(n,d[:,i].mean(),*np.percentile(d[:,i],[10,50,90])))
tot=a[:,6]-a[:,0]
print('WG total              mean %8.0f  p50 %8.0f'%(tot.mean(),np.median(tot)))
# gaps between consecutive WGs on the same CU are unknown; estimate concurrency = sum(WG total)/span
print('avg concurrent WGs:', tot.sum()/(a[:,6].max()-t0))
# start-time distribution
st=np.sort(a[:,0]-t0); print('WG start times: first 5', st[:5], ' ... 256th', st[min(255,nb-1)], ' 512th', st[min(511,nb-1)], 'last', st[-1])

# per-CU timelines: group by (xcc, se/sh/cu bits of HW_ID)
ids=a[:,7]
xcc=(ids>>32)&0xf
hw=ids&0xffffffff
cu=(hw>>8)&0xf; sh=(hw>>12)&1; se=(hw>>13)&7
key=xcc*1000+se*100+sh*50+cu
import collections
gaps=[];busy=[]
for k in np.unique(key):
    m=key==k
    st=a[m,0]; en=a[m,6]
    o=np.argsort(st); st=st[o]; en=en[o]
    if len(st)>1:
        gaps.extend((st[1:]-en[:-1]).tolist())
    busy.append((en-st).sum()/(en.max()-st.min()))
gaps=np.array(gaps)
print('distinct CUs seen:',len(np.unique(key)),' WGs per CU: %.1f'%(nb/len(np.unique(key))))
print('gap between consecutive WGs on a CU: mean %.0f p10 %.0f p50 %.0f p90 %.0f (negative = overlapping WGs)'%(gaps.mean(),*np.percentile(gaps,[10,50,90])))
print('per-CU busy fraction mean %.3f'%np.mean(busy))
span=[a[key==k,6].max()-a[key==k,0].min() for k in np.unique(key)]
print('per-CU span ticks mean %.0f max %.0f'%(np.mean(span),np.max(span)))
