"""Container walkers of geotransformer/utils/torch.py:97-122 (`to_cuda` / `release_cuda`) + its seeding helper (:83-94): what
experiments/*/demo.py:59-62 and test.py wrap around the model call.  Device = the current HIP device."""
import random

import numpy as np
import torch


def _walk(x, leaf):
    if isinstance(x, list):
        return [_walk(v, leaf) for v in x]
    if isinstance(x, tuple):
        return tuple(_walk(v, leaf) for v in x)
    if isinstance(x, dict):
        return {k: _walk(v, leaf) for k, v in x.items()}
    return leaf(x) if isinstance(x, torch.Tensor) else x


def to_cuda(x):
    """Every tensor of a nested list / tuple / dict moved to the device; everything else untouched."""
    return _walk(x, lambda t: t.cuda())


def release_cuda(x):
    """Every tensor of a nested container -> python scalar (one element) or numpy array, as the reference returns them."""
    return _walk(x, lambda t: t.item() if t.numel() == 1 else t.detach().cpu().numpy())


def initialize(seed=None, cudnn_deterministic=True, autograd_anomaly_detection=False):
    """Seeds python / numpy / torch like the reference; the MIOpen switches are irrelevant here (no library convolutions on the path)."""
    del cudnn_deterministic
    if seed is not None:
        random.seed(seed)
        torch.manual_seed(seed)
        np.random.seed(seed)
    torch.autograd.set_detect_anomaly(autograd_anomaly_detection)
