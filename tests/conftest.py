import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Golden:
    """npz fixture with '/'-separated keys; json strings decoded on access via .json(key)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'))

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z.files

    def json(self, k):
        return json.loads(str(self.z[k]))

    def sub(self, prefix):
        """dict of arrays whose key starts with prefix (prefix stripped)."""
        return {k[len(prefix):]: self.z[k] for k in self.z.files if k.startswith(prefix)}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


@pytest.fixture(scope='session', autouse=True)
def _orderly_gpu_teardown():
    """Drop captured HIP graphs / cached tensors and drain the device before the interpreter starts tearing modules down."""
    yield
    import gc
    gc.collect()
    if 'torch' in sys.modules:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
