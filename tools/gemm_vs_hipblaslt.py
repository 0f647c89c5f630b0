import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laff_amd import ops
dev='cuda'
Nt,Nv,K=16384,16384,4096
t=torch.nn.functional.normalize(torch.randn(Nt,K,device=dev),dim=1); v=torch.nn.functional.normalize(torch.randn(Nv,K,device=dev),dim=1)
T=ops.pack_rows(t,True,1e-13,'fp16'); V=ops.pack_rows(v,True,1e-13,'fp16')
S=torch.empty(Nt,Nv,device=dev)
a16=t.half(); b16=v.half()
def timed(fn, secs, label):
    fn(); torch.cuda.synchronize()
    stop=[False]; out=[]
    def poll():
        while not stop[0]:
            try:
                r=subprocess.run(['rocm-smi','--showclocks','--showpower'],capture_output=True,text=True,timeout=10).stdout
                out.append(' | '.join(l.strip() for l in r.splitlines() if ('sclk' in l or 'Power' in l or 'fclk' in l or 'mclk' in l)))
            except Exception as e: out.append(repr(e))
            time.sleep(0.4)
    th=threading.Thread(target=poll); th.start()
    n=0; e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    t0=time.time(); e0.record()
    while time.time()-t0<secs:
        for _ in range(20): fn()
        n+=20; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0]=True; th.join()
    ms=e0.elapsed_time(e1)/n
    print(label,'ms %.4f TF %.1f'%(ms,2.0*Nt*Nv*K/ms/1e9))
    for o in out[:6]: print('   ',o)
# like for like: both libraries with fp32 output (1.07 GB written) and without it (hipBLASLt: fp16 output, 0.54 GB; laff: count-only --
# the banded rank count in the epilogue, no score matrix)
timed(lambda: ops.sim_gemm(T,V,out=S), 3.0, 'laff sim_gemm fp16 16384x16384x4096 (fp32 out)')
try:
    C32=torch.empty(Nt,Nv,device=dev,dtype=torch.float32)
    timed(lambda: torch.mm(a16,b16.t(),out_dtype=torch.float32,out=C32), 3.0, 'torch.mm fp16 (hipBLASLt) fp32 out (out_dtype)')
    del C32
except Exception as e:
    print('torch.mm out_dtype=float32 not available:', repr(e)[:200])
C=torch.empty(Nt,Nv,device=dev,dtype=torch.float16)
timed(lambda: torch.matmul(a16,b16.t(),out=C), 3.0, 'torch.matmul fp16 (hipBLASLt) fp16 out')
gt=(torch.arange(Nt,device=dev)%Nv).int()
# a planted match per text (as tools/debug/time_shape.py): the ground-truth score stands clear of the bulk, a realistic share of pairs inside the band
tp=torch.nn.functional.normalize(t+0.5*v[gt.long()],dim=1)
Tp=ops.pack_rows(tp,True,1e-13,'fp16')
st=ops.rank_prepare(tp.reshape(Nt,1,K),v.reshape(Nv,1,K),Tp,V,gt)
def count_only():
    st.pairs[:4].zero_(); st.count.zero_()
    ops.sim_gemm_banded(st, want_scores=False)
timed(count_only, 3.0, 'laff sim_gemm_banded fp16 16384x16384x4096 count-only (no output matrix; + two one-block clears per launch)')
del st
a=torch.randn(40000,512,device=dev).half(); b=torch.randn(10000,512,device=dev).half()
Nt,Nv,K=40000,10000,512
C2=torch.empty(40000,10000,device=dev,dtype=torch.float16)
timed(lambda: torch.matmul(a,b.t(),out=C2), 2.0, 'torch.matmul fp16 40000x10000x512 fp16 out')
T2=ops.pack_rows(a.float(),True,1e-13,'fp16'); V2=ops.pack_rows(b.float(),True,1e-13,'fp16'); S2=torch.empty(40000,10000,device=dev)
timed(lambda: ops.sim_gemm(T2,V2,out=S2), 2.0, 'laff sim_gemm fp16 40000x10000x512 fp32 out')
