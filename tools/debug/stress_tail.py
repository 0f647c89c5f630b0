#!/usr/bin/env python3
"""Stress of the fence-free metrics tails (rank.hip): many graph replays of {prepare, banded GEMM, tail} on a large grid; prints one JSON line
with the distinct metric tuples seen.  Run once with the shipped library and once with a -DLAFF_TAIL_FENCES build (LAFF_HIP_LIB): the
test test_fence_free_tail_equals_the_fenced_build compares the two.
usage: stress_tail.py [replays] [fused|split]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from laff_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fused = (sys.argv[2] if len(sys.argv) > 2 else 'fused') == 'fused'
dev = torch.device('cuda')
g = torch.Generator(device='cuda').manual_seed(5)
Nt, Nv, d = (16384 if fused else 40000), 6144, 512
z = torch.randn(Nv, 48, generator=g, device=dev); P = torch.randn(48, d, generator=g, device=dev)
gt = (torch.arange(Nt, device=dev) * 7919 % Nv).to(torch.int32)
Ev = (z @ P + 0.9 * torch.randn(Nv, d, generator=g, device=dev)).reshape(Nv, 1, d).contiguous()
Et = (z[gt.long()] @ P + 0.9 * torch.randn(Nt, d, generator=g, device=dev)).reshape(Nt, 1, d).contiguous()
Et = torch.nn.functional.normalize(Et, dim=2); Ev = torch.nn.functional.normalize(Ev, dim=2)
T = ops.pack_rows(Et, False, 1e-13, 'fp16'); V = ops.pack_rows(Ev, False, 1e-13, 'fp16')
ops.ctx_prepare_metrics(dev)
pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
ranks = torch.empty(Nt, dtype=torch.int32, device=dev)
def step():
    st = ops.rank_prepare(Et, Ev, T, V, gt)
    ops.sim_gemm_banded(st, want_scores=False)
    if fused:
        ops.rank_resolve_metrics(st, None, pinned, ranks_out=ranks)
    else:
        ops.rank_resolve(st, None)
        ops.rank_metrics_async(st.count, pinned, base=1, ranks_out=ranks)
    return st
step(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, capture_error_mode='thread_local'):
    step()
seen = {}
r0 = None
for i in range(reps):
    pinned.fill_(-1.0)
    gr.replay()
    torch.cuda.synchronize()
    key = tuple(pinned.tolist())
    seen[key] = seen.get(key, 0) + 1
    if r0 is None:
        r0 = ranks.clone()
    elif not torch.equal(r0, ranks):
        seen[('ranks differ', i)] = 1
print(json.dumps({'replays': reps, 'fused': fused, 'lib': os.environ.get('LAFF_HIP_LIB', 'default'),
                  'distinct': [[list(map(str, k)), v] for k, v in seen.items()], 'rank_sum': int(r0.long().sum())}))
