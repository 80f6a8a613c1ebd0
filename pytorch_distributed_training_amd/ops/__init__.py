"""Op backends.

``ref``  pure-PyTorch fp32 reference (numerical oracle; the CPU / gloo path).
``hip``  hand-written gfx950 HIP kernels (the MI355X path).

There is no automatic fall-back from ``hip`` to ``ref`` on a GPU: if the compiled
extension cannot be loaded, selecting a backend for a GPU device raises.
"""
from __future__ import annotations

import torch

from . import ref


def get_ops(device: torch.device, backend: str = "auto"):
    """Return the op backend module for ``device`` (``auto``: hip on GPU, ref on CPU)."""
    device = torch.device(device)
    if backend == "auto":
        backend = "hip" if device.type == "cuda" else "ref"
    if backend == "ref":
        return ref
    if backend == "hip":
        if device.type != "cuda":
            raise RuntimeError("the hip op backend needs a GPU device")
        from . import hip  # raises with a clear message when the extension is not built
        return hip
    raise ValueError(f"unknown op backend '{backend}' (auto | hip | ref)")


__all__ = ["get_ops", "ref"]
