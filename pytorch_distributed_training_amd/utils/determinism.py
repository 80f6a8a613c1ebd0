"""Seeding helpers.

Upstream calls ``make_deterministic(seed)`` once in the launcher and once more in every
worker with the *same* seed, so that all ranks build identical initial weights
(reference ``train_distributed.py:51-53,141-142``).
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch


def make_deterministic(seed: int, strict: bool = False) -> None:
    """Seed python, numpy and torch (CPU and every visible GPU) RNGs.

    ``strict=True`` (``runtime.deterministic``) additionally asks torch for deterministic
    algorithms; the same config key makes the native engine switch its HIP kernels to ordered
    reductions (``ops.hip.set_deterministic``), so two GPU runs with one seed end with the same
    parameter bits.  Upstream is not bit-deterministic: it enables cudnn autotuning right after
    seeding (``train_distributed.py:54``).
    """
    seed = int(seed)
    os.environ.setdefault("PYTHONHASHSEED", str(seed))
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.device_count() > 0 and torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if strict:
        torch.use_deterministic_algorithms(True, warn_only=True)
