"""Flat parameter / gradient / momentum storage.

Upstream keeps one tensor per parameter, lets autograd allocate one gradient tensor per
parameter and lets the DDP reducer copy every gradient into flat buckets and back
(reference ``train_distributed.py:198,207,276-278``; bucket copies in
``torch/include/torch/csrc/distributed/c10d/reducer.hpp:275,354-363``).  Here all
trainable parameters live in ONE fp32 buffer, with a same-shaped gradient buffer, momentum
buffer and (on GPU) a bf16 compute mirror:

* weight-gradient kernels accumulate straight into the gradient buffer, whose element
  order is the order in which backward produces gradients, so a gradient bucket is a
  contiguous slice and needs no copy-in / copy-out;
* the optimizer is one fused kernel over the flat buffers (it also zeroes the gradients
  and refreshes the bf16 mirror);
* convolution weights are stored OHWI (``[Cout, R, S, Cin]``), the layout the NHWC
  implicit-GEMM kernels consume.  ``state_dict`` import/export converts to and from the
  torchvision layout (OIHW, fp32) so checkpoints stay interchangeable.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Sequence, Tuple

import torch

_ALIGN = 64  # elements; keeps every entry 256-byte (fp32) / 128-byte (bf16) aligned


@dataclass
class ParamEntry:
    key: str
    shape_tv: Tuple[int, ...]        # torchvision (state_dict) shape
    shape: Tuple[int, ...]           # internal shape (OHWI for 4-D weights)
    offset: int
    numel: int
    is_conv: bool
    no_decay: bool                   # BN weights/biases and the classifier bias


def _internal_shape(shape_tv: Sequence[int]) -> Tuple[int, ...]:
    if len(shape_tv) == 4:
        o, i, r, s = shape_tv
        return (o, r, s, i)
    return tuple(shape_tv)


class FlatParams:
    """Owns the flat buffers; hands out named views."""

    def __init__(self, ordered: Iterable[Tuple[str, Tuple[int, ...]]], device: torch.device,
                 compute_dtype: torch.dtype) -> None:
        self.device = device
        self.compute_dtype = compute_dtype
        self.entries: Dict[str, ParamEntry] = {}
        self.order: List[str] = []
        off = 0
        for key, shape_tv in ordered:
            shape = _internal_shape(shape_tv)
            n = 1
            for d in shape:
                n *= d
            no_decay = len(shape_tv) == 1
            self.entries[key] = ParamEntry(key, tuple(shape_tv), shape, off, n, len(shape_tv) == 4, no_decay)
            self.order.append(key)
            off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        if compute_dtype == torch.float32:
            self.pc = self.p                      # no mirror needed
        else:
            self.pc = torch.zeros(off, dtype=compute_dtype, device=device)

    # -- views ------------------------------------------------------------------------
    def _view(self, buf: torch.Tensor, key: str) -> torch.Tensor:
        e = self.entries[key]
        return buf[e.offset:e.offset + e.numel].view(e.shape)

    def param(self, key: str) -> torch.Tensor:
        return self._view(self.p, key)

    def grad(self, key: str) -> torch.Tensor:
        return self._view(self.g, key)

    def momentum(self, key: str) -> torch.Tensor:
        return self._view(self.m, key)

    def compute(self, key: str) -> torch.Tensor:
        return self._view(self.pc, key)

    def end_offset(self, key: str) -> int:
        """Flat offset just past ``key`` (padding included) -- used to mark buckets ready."""
        idx = self.order.index(key)
        if idx + 1 < len(self.order):
            return self.entries[self.order[idx + 1]].offset
        return self.total

    def num_parameters(self) -> int:
        return sum(e.numel for e in self.entries.values())

    def refresh_compute(self) -> None:
        if self.pc is not self.p:
            self.pc.copy_(self.p)

    # -- layout conversion ------------------------------------------------------------
    @staticmethod
    def to_internal(t: torch.Tensor) -> torch.Tensor:
        return t.permute(0, 2, 3, 1) if t.dim() == 4 else t

    @staticmethod
    def to_torchvision(t: torch.Tensor) -> torch.Tensor:
        return t.permute(0, 3, 1, 2) if t.dim() == 4 else t

    def load_from(self, state: Dict[str, torch.Tensor], which: str = "p", strict: bool = True) -> None:
        """Copy torchvision-layout tensors into buffer ``which`` ('p' or 'm')."""
        buf = {"p": self.p, "m": self.m, "g": self.g}[which]
        for key, e in self.entries.items():
            if key not in state:
                if strict:
                    raise KeyError(f"missing parameter '{key}' in state dict")
                continue
            src = state[key]
            if tuple(src.shape) != e.shape_tv:
                raise ValueError(f"shape mismatch for '{key}': got {tuple(src.shape)}, expected {e.shape_tv}")
            self._view(buf, key).copy_(self.to_internal(src.detach().to(torch.float32)))
        if which == "p":
            self.refresh_compute()

    def export(self, which: str = "p") -> Dict[str, torch.Tensor]:
        """Torchvision-layout fp32 contiguous CPU-or-device clones, in registration order."""
        buf = {"p": self.p, "m": self.m, "g": self.g}[which]
        out = {}
        for key in self.order:
            out[key] = self.to_torchvision(self._view(buf, key)).contiguous().clone()
        return out


def make_buckets(params: FlatParams, cap_mb: float, first_mb: float, tail_mb: float = 2.0) -> List[Tuple[int, int]]:
    """Split the flat gradient buffer into contiguous ``[start, end)`` element ranges.

    The first bucket (the gradients backward produces first) is small so communication
    starts early; the rest are capped at ``cap_mb``.  The LAST bucket cannot overlap anything
    (it completes with the final weight gradient, its all-reduce is exposed), so it is cut down
    to the final ``tail_mb``: in a ResNet the early layers, whose gradients come last, hold few
    parameters, and what precedes them in a cap-sized last bucket was ready milliseconds
    earlier.  Entries are never split.
    """
    cap = max(int(cap_mb * 2 ** 20 / 4), 1)
    first = max(int(first_mb * 2 ** 20 / 4), 1)
    buckets: List[Tuple[int, int]] = []
    start = 0
    limit = first
    for key in params.order:
        end = params.end_offset(key)
        if end - start >= limit:
            buckets.append((start, end))
            start = end
            limit = cap
    if start < params.total:
        buckets.append((start, params.total))
    tail = int(tail_mb * 2 ** 20 / 4)
    if tail > 0 and buckets and buckets[-1][1] - buckets[-1][0] > 2 * tail:
        lo, hi = buckets[-1]
        cut = None
        for key in params.order:                    # first entry boundary inside the last `tail` elements
            off = params.entries[key].offset
            if lo < off and hi - off <= tail:
                cut = off
                break
        if cut is not None:
            buckets[-1] = (lo, cut)
            buckets.append((cut, hi))
    return buckets
