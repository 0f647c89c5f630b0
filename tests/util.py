"""Shared helpers for the parity tests: build laff_amd models that mirror the golden fixtures."""
import numpy as np
import torch


def load_sd(model, sd_arrays, device=None):
    """load a fixture state_dict (numpy arrays) with strict=False like predictor.py:167; returns missing/unexpected."""
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_arrays.items()}
    res = model.load_state_dict(sd, strict=False)
    return res


def maxdiff(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
