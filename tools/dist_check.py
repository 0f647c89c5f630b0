#!/usr/bin/env python3
"""Multi-GPU self-check of laff_amd.dist (run under torchrun, one process per GPU):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dist_check.py

Every rank builds the same synthetic model and features (seeded), evaluates its shards with both decompositions ('video' and
'text'), eagerly and through per-phase HIP graphs (GraphRunner), on a problem whose shards are UNEVEN, and compares the ranks and
the 7 metrics with a single-GPU pass of the whole problem computed on every rank.  Exit code 0 = identical everywhere.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from laff_amd import retrieval, synth
    from laff_amd.dist import GraphRunner, HipBackend, check_metrics_flag, evaluate_sharded, evaluate_sharded_by_text, shard_bounds
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    Nt, Nv, H, d = 10007, 3001, 1, 512                     # uneven for every world size 2..8
    model = synth.build_model(H, d, dev)
    vis, txt, gt, _ = synth.make_features(Nt, Nv, dev)
    ref = retrieval.evaluate(model, vis, txt, gt, precision='fp16')
    t0, t1 = shard_bounds(Nt, world, rank)
    v0, v1 = shard_bounds(Nv, world, rank)
    vis_l = {k: synth.slice_rows(v, v0, v1) for k, v in vis.items()}
    txt_l = {k: synth.slice_rows(v, t0, t1) for k, v in txt.items()}
    backend = HipBackend(model, 'fp16')
    bad = 0
    for name, fn in (('video', evaluate_sharded), ('text', evaluate_sharded_by_text)):
        res = fn(backend, vis_l, txt_l, gt, Nt, Nv, H)
        ok = torch.equal(res['ranks'][:Nt], ref.ranks) and max(abs(a - b) for a, b in zip(res['metrics'], ref.metrics)) < 1e-9
        runner, state = GraphRunner(), {}
        pins = [torch.zeros(8, dtype=torch.float64).pin_memory() for _ in range(2)]
        for it in range(4):                                  # capture (x2 finish tags), then replays
            out = fn(backend, vis_l, txt_l, gt, Nt, Nv, H, runner=runner, state=state, metrics_out=pins[it % 2],
                     finish_tag=str(it % 2))
            torch.cuda.synchronize()
            check_metrics_flag(pins[it % 2])
            ok = ok and torch.equal(out['ranks'][:Nt], ref.ranks)
            ok = ok and max(abs(float(a) - b) for a, b in zip(pins[it % 2][:7], ref.metrics)) < 1e-9
        print('rank %d/%d  scheme %-5s  %s' % (rank, world, name, 'OK' if ok else 'MISMATCH'), flush=True)
        bad += 0 if ok else 1
    flag = torch.tensor([bad], device=dev)
    dist.all_reduce(flag)
    dist.destroy_process_group()
    sys.exit(1 if int(flag.item()) else 0)


if __name__ == '__main__':
    main()
