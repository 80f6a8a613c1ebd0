"""Bucketed gradient all-reduce overlapped with backward.

Replaces the ``DistributedDataParallel`` C++ ``Reducer`` the upstream trainer relies on
(reference ``train_distributed.py:198``; bucketing in
``torch/nn/parallel/distributed.py:1199-1251``, 1 MiB first bucket + 25 MiB caps,
``torch/include/torch/csrc/distributed/c10d/reducer.hpp:30-31``).

Design for an xGMI mesh of MI355X GPUs:

* gradients already live in one flat fp32 buffer in backward-production order
  (:mod:`..engine.params`), so a bucket is a contiguous slice -- there is no copy into or
  out of bucket storage and no per-parameter hook;
* the engine calls :meth:`GradReducer.mark_ready` with the flat offset up to which
  gradients are final; every bucket that became complete is all-reduced (SUM, in place)
  over RCCL on a dedicated communication stream that waits on an event recorded on the
  compute stream, so communication of bucket *k* overlaps the backward kernels of the
  layers feeding bucket *k+1*;
* the ``1/world_size`` average is not applied here: the fused optimizer kernel folds it
  into its gradient read (``grad_scale``);
* on CPU (gloo) the same code path issues asynchronous all-reduces and waits on the
  returned work handles.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, buckets: List[Tuple[int, int]], group=None,
                 overlap: bool = True, compress_bf16: bool = False) -> None:
        self.flat = flat_grad
        self.buckets = list(buckets)
        self.group = group
        initialized = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if initialized else 1
        # PDT_FORCE_COLLECTIVES=1 issues the collectives even in a one-rank group (debug: drives
        # the RCCL / stream / event path on a single GPU).
        self.enabled = self.world > 1 or (initialized and os.environ.get("PDT_FORCE_COLLECTIVES") == "1")
        self.overlap = overlap
        self.compress_bf16 = compress_bf16
        self.is_cuda = flat_grad.is_cuda
        self._next = 0
        self._pending = []
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self._ready_events: List = []
        self._done_events: List = []
        if self.enabled and self.is_cuda:
            self._comm_stream = torch.cuda.Stream(device=flat_grad.device) if overlap else None
            self._ready_events = [torch.cuda.Event() for _ in self.buckets]
            self._done_events = [torch.cuda.Event() for _ in self.buckets]
        self.stats = {"launched": 0, "bytes": 0}

    # ---------------------------------------------------------------------------------
    def begin_step(self) -> None:
        self._next = 0
        self._pending.clear()

    def mark_ready(self, end_offset: int) -> None:
        """Gradients in ``flat[:end_offset]`` are final: launch every completed bucket."""
        if not self.enabled:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][1] <= end_offset:
            self._launch(self._next)
            self._next += 1

    def completes_bucket(self, end_offset: int) -> bool:
        """Would :meth:`mark_ready` launch anything?  Lets a caller that records host actions
        (step-graph capture) skip the offsets that do not finish a bucket."""
        return self.enabled and self._next < len(self.buckets) and self.buckets[self._next][1] <= end_offset

    def _launch(self, idx: int) -> None:
        start, end = self.buckets[idx]
        view = self.flat[start:end]
        self.stats["launched"] += 1
        self.stats["bytes"] += view.numel() * 4
        if self.is_cuda and self._comm_stream is not None:
            cur = torch.cuda.current_stream(self.flat.device)
            self._ready_events[idx].record(cur)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(self._ready_events[idx])
                self._all_reduce(view)
                self._done_events[idx].record(self._comm_stream)
            self._pending.append(idx)
        elif self.is_cuda:
            self._all_reduce(view)
        else:
            work = dist.all_reduce(view, group=self.group, async_op=True)
            self._pending.append(work)

    def _all_reduce(self, view: torch.Tensor) -> None:
        if self.compress_bf16:
            tmp = view.to(torch.bfloat16)
            dist.all_reduce(tmp, group=self.group)
            view.copy_(tmp)
        else:
            dist.all_reduce(view, group=self.group)

    def finish(self) -> None:
        """Launch any remaining bucket and make the compute stream wait for all of them."""
        if not self.enabled:
            return
        self.mark_ready(self.flat.numel())
        if self.is_cuda and self._comm_stream is not None:
            cur = torch.cuda.current_stream(self.flat.device)
            for idx in self._pending:
                cur.wait_event(self._done_events[idx])
        elif not self.is_cuda:
            for work in self._pending:
                work.wait()
        self._pending.clear()

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world
