"""The helpers every reference example starts with (dance/utils/__init__.py:19-95): device resolution and global seeding."""
import hashlib
import os
import random
from typing import Any

import numpy as np
import torch


def get_device(device: str) -> str:
    """``"auto"`` -> the GPU (dance/utils/__init__.py:19-22; there is no CPU path behind it here)."""
    return "cuda" if device == "auto" else device


def hexdigest(x: str, /) -> str:
    return hashlib.md5(x.encode()).hexdigest()


def default(value: Any, default_value: Any):
    return default_value if value is None else value


def set_seed(rndseed, cuda: bool = True, extreme_mode: bool = False):
    """Seed python, numpy and torch (host and every visible GPU), as dance/utils/__init__.py:81-94 does; the mini-batch loaders and
    the samplers of this package draw from torch's generators, so this fixes their order too (the reference also seeds dgl).
    ``extreme_mode`` asks torch for deterministic library kernels; the HIP kernels of this package are deterministic already."""
    os.environ["PYTHONHASHSEED"] = str(rndseed)
    random.seed(rndseed)
    np.random.seed(rndseed)
    torch.manual_seed(rndseed)
    if cuda and torch.cuda.is_available():
        torch.cuda.manual_seed_all(rndseed)
    if extreme_mode:
        torch.backends.cudnn.benchmark = False
        torch.backends.cudnn.deterministic = True
