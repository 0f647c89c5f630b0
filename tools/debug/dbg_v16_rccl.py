"""1-rank RCCL run of dist.evaluate_sharded_v16 with a mark after every phase (debug aid)."""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from laff_amd import synth
from laff_amd.dist import GraphRunner, HipBackend, evaluate_sharded, evaluate_sharded_v16


class T:
    def __init__(self): self.t = time.time()
    def mark(self, n):
        torch.cuda.synchronize(); print('  mark %-20s %.3f s' % (n, time.time() - self.t), flush=True)


s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
dev = torch.device('cuda')
Nt, Nv, H, d, _ = synth.WORKLOADS['c2_10kx3k']
model = synth.build_model(H, d, dev)
vis, txt, gt, _ = synth.make_features(Nt, Nv, dev)
backend = HipBackend(model, 'fp16')
ref = evaluate_sharded(backend, vis, txt, gt, Nt, Nv, H)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, world_size=1, rank=0, device_id=torch.device('cuda', torch.cuda.current_device()))
print('eager', flush=True)
out = evaluate_sharded_v16(backend, vis, txt, gt, Nt, Nv, H, force_collectives=True, timer=T())
print('eager ok', torch.equal(out['ranks'], ref['ranks']), flush=True)
pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
class PartRunner(GraphRunner):
    def __call__(self, name, fn):
        if name in os.environ.get('EAGER_PHASES', '').split(','):
            return fn()
        return super().__call__(name, fn)


runner, state = PartRunner(), {}
for i in range(3):
    print('graph pass', i, flush=True)
    out = evaluate_sharded_v16(backend, vis, txt, gt, Nt, Nv, H, force_collectives=True, runner=runner, state=state, metrics_out=pinned, timer=T())
    torch.cuda.synchronize()
    print('  ok', torch.equal(out['ranks'], ref['ranks']), pinned.tolist(), flush=True)
dist.destroy_process_group()
