"""Which limiter holds the clock?  Runs one workload at a time for a few seconds and samples `amd-smi metric -p -c -t -v` (power, clocks,
temperatures, violation / throttle status) from a side thread: the package cap (ppt), a thermal limit, or neither.
    python tools/debug/throttle_reason.py [outfile]
Prints one block per workload: rate, then the sampled fields that differ from idle."""
import sys, os, subprocess, threading, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from laff_amd import ops
dev = 'cuda'
out_path = sys.argv[1] if len(sys.argv) > 1 else None
log = open(out_path, 'w') if out_path else None

def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    if log: log.write(s + '\n'); log.flush()

def smi():
    try:
        r = subprocess.run(['amd-smi', 'metric', '-g', '0', '-p', '-c', '-t', '-v', '--json'], capture_output=True, text=True, timeout=20)
        return json.loads(r.stdout)
    except Exception as e:
        return {'error': repr(e)[:200]}

def flat(d, pre=''):
    o = {}
    if isinstance(d, dict):
        for k, v in d.items(): o.update(flat(v, pre + str(k) + '.'))
    elif isinstance(d, list):
        for i, v in enumerate(d): o.update(flat(v, pre + str(i) + '.'))
    else:
        o[pre[:-1]] = d
    return o

idle = flat(smi())
say('== idle'); [say('   ', k, '=', v) for k, v in idle.items()]

def timed(fn, secs, label, flops):
    fn(); torch.cuda.synchronize()
    stop = [False]; samples = []
    def poll():
        time.sleep(0.8)
        while not stop[0]:
            samples.append(flat(smi())); time.sleep(0.3)
    th = threading.Thread(target=poll); th.start()
    n = 0; e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.time(); e0.record()
    while time.time() - t0 < secs:
        for _ in range(40): fn()
        n += 40; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop[0] = True; th.join()
    ms = e0.elapsed_time(e1) / n
    say('== %s: %.4f ms  %.1f TF' % (label, ms, flops / ms / 1e9))
    if not samples: return
    keys = [k for k in samples[-1] if any(s.get(k) != idle.get(k) for s in samples)]
    for k in keys:
        vals = [s.get(k) for s in samples]
        if all(isinstance(v, (int, float)) for v in vals) and len(set(vals)) > 1:
            say('    %s: first %s last %s (idle %s)' % (k, vals[0], vals[-1], idle.get(k)))
        else:
            say('    %s: %s (idle %s)' % (k, vals[-1], idle.get(k)))

Nt, Nv, K = 16384, 16384, 4096
t = torch.nn.functional.normalize(torch.randn(Nt, K, device=dev), dim=1); v = torch.nn.functional.normalize(torch.randn(Nv, K, device=dev), dim=1)
T = ops.pack_rows(t, True, 1e-13, 'fp16'); V = ops.pack_rows(v, True, 1e-13, 'fp16')
S = torch.empty(Nt, Nv, device=dev)
a16 = t.half(); b16 = v.half()
fl = 2.0 * Nt * Nv * K
timed(lambda: ops.sim_gemm(T, V, out=S), 4.0, 'laff sim_gemm fp16 16384^2 x 4096 fp32 out', fl)
C32 = torch.empty(Nt, Nv, device=dev, dtype=torch.float32)
timed(lambda: torch.mm(a16, b16.t(), out_dtype=torch.float32, out=C32), 4.0, 'hipBLASLt fp16 16384^2 x 4096 fp32 out', fl)
del C32
C = torch.empty(Nt, Nv, device=dev, dtype=torch.float16)
timed(lambda: torch.matmul(a16, b16.t(), out=C), 4.0, 'hipBLASLt fp16 16384^2 x 4096 fp16 out', fl)
del C, S, T, V, t, v, a16, b16
a = torch.randn(40000, 512, device=dev); b = torch.randn(10000, 512, device=dev)
T2 = ops.pack_rows(a, True, 1e-13, 'fp16'); V2 = ops.pack_rows(b, True, 1e-13, 'fp16'); S2 = torch.empty(40000, 10000, device=dev)
timed(lambda: ops.sim_gemm(T2, V2, out=S2), 4.0, 'laff sim_gemm fp16 40000 x 10000 x 512 fp32 out (strip kernel)', 2.0 * 40000 * 10000 * 512)
