"""N > 1 processes over RCCL on real GPUs: runs only where a node with at least two MI355X is visible (self-skips on the 1-GPU
boxes of this pool; the orchestration itself is covered on CPU by tests/test_dist_gloo.py and on one GPU by the 1-rank RCCL tests)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world', [2, 3, 8])
def test_sharded_ranks_equal_single_gpu_over_rccl(world):
    import torch
    if torch.cuda.device_count() < world:            # (device_count does not initialise the GPU in this process)
        pytest.skip('needs %d GPUs, found %d' % (world, torch.cuda.device_count()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                          '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'tools', 'dist_check.py')],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert out.stdout.count('OK') == 2 * world and 'MISMATCH' not in out.stdout
