"""Bucketed gradient all-reduce overlapped with backward, with a per-bucket follow-up.

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
* ``bucket_hook(start, end)`` (optional) is enqueued on the same stream right behind the
  bucket's all-reduce: the engine puts the fused optimizer update of that slice there, so
  when backward ends only the last (small) bucket and its slice of the update are still to
  run -- the whole-model optimizer pass no longer sits serially behind backward.  With a hook
  the pipeline also runs in a one-rank job (no collective, only the hook on the side stream);
* gradients written on a second compute stream (the engine's weight-gradient stream) are waited for by the
  communication stream, not by the main compute stream (:meth:`GradReducer.add_producer_stream`);
* the ``1/world_size`` average is not applied here: the fused optimizer kernel folds it
  into its gradient read (``grad_scale``);
* ``comm_dtype=torch.bfloat16`` halves the bytes on the links: the bucket is rounded to bf16,
  summed, and written back (``training.grad_allreduce_dtype``);
* on CPU (gloo) the same code path issues asynchronous all-reduces and waits on the
  returned work handles (hooks run in bucket order after the waits).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import collectives as coll
from . import streams


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, buckets: List[Tuple[int, int]], group=None,
                 overlap: bool = True, comm_dtype: Optional[torch.dtype] = None) -> None:
        self.flat = flat_grad
        self.buckets = list(buckets)
        self.group = group
        initialized = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if initialized else 1
        # PDT_FORCE_COLLECTIVES=1 issues the collectives even in a one-rank group (debug: drives
        # the RCCL / stream / event path on a single GPU).
        self.collective = self.world > 1 or (initialized and os.environ.get("PDT_FORCE_COLLECTIVES") == "1")
        self.overlap = overlap
        self.comm_dtype = comm_dtype if comm_dtype not in (None, torch.float32) else None
        self.is_cuda = flat_grad.is_cuda
        self.bucket_hook: Optional[Callable[[int, int], None]] = None
        self._next = 0
        self._pending: list = []
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self._ready_events: List = []
        self._done_events: List = []
        self._timing: Optional[list] = None
        self._producers: list = []            # streams besides the current one that write gradients (see below)
        self._producer_events: List[list] = []
        self.stats = {"launched": 0, "bytes": 0, "steps": 0}

    # ---------------------------------------------------------------------------------
    @property
    def enabled(self) -> bool:
        """Is there anything to do per bucket (a collective, a hook, or both)?"""
        return self.collective or self.bucket_hook is not None

    def set_bucket_hook(self, hook: Optional[Callable[[int, int], None]]) -> None:
        self.bucket_hook = hook

    def add_producer_stream(self, stream) -> bool:
        """Gradients are also written by kernels on ``stream`` (the engine's weight-gradient stream).  With the
        overlapped GPU path the COMMUNICATION stream then waits for that stream at every bucket, so the compute
        stream never has to (a join there stalls the critical path until the weight gradients have caught up).
        Returns False when this reducer cannot take the dependency over (the caller keeps joining itself)."""
        if not (self.is_cuda and self.overlap):
            return False
        self._producers.append(stream)
        return True

    def _ensure_streams(self) -> None:
        if self.is_cuda and self.overlap and self._comm_stream is None:
            self._comm_stream = streams.get("comm", self.flat.device)
            self._ready_events = [torch.cuda.Event() for _ in self.buckets]
            self._done_events = [torch.cuda.Event() for _ in self.buckets]
        if self.is_cuda and self.overlap and len(self._producer_events) != len(self.buckets):
            self._producer_events = [[torch.cuda.Event() for _ in self._producers] for _ in self.buckets]

    def begin_step(self) -> None:
        self._next = 0
        self._pending.clear()
        self.stats["steps"] += 1

    def mark_ready(self, end_offset: int) -> None:
        """Gradients in ``flat[:end_offset]`` are final: launch every completed bucket."""
        if not self.enabled:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][1] <= end_offset:
            self._launch(self._next)
            self._next += 1

    def completes_bucket(self, end_offset: int) -> bool:
        """Would :meth:`mark_ready` launch anything?  Lets a caller that records host actions
        (step replay) skip the offsets that do not finish a bucket."""
        return self.enabled and self._next < len(self.buckets) and self.buckets[self._next][1] <= end_offset

    def _launch(self, idx: int) -> None:
        start, end = self.buckets[idx]
        view = self.flat[start:end]
        if self.collective:
            self.stats["launched"] += 1
            self.stats["bytes"] += view.numel() * (2 if self.comm_dtype is torch.bfloat16 else 4)
        if self.is_cuda and self.overlap:
            self._ensure_streams()
            cur = torch.cuda.current_stream(self.flat.device)
            self._ready_events[idx].record(cur)
            for ev, producer in zip(self._producer_events[idx], self._producers):
                ev.record(producer)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(self._ready_events[idx])
                for ev in self._producer_events[idx]:
                    self._comm_stream.wait_event(ev)
                if self.collective:
                    self._all_reduce(view)
                if self.bucket_hook is not None:
                    self.bucket_hook(start, end)
                self._done_events[idx].record(self._comm_stream)
            self._pending.append(idx)
        elif self.is_cuda:
            if self.collective:
                self._all_reduce(view)
            if self.bucket_hook is not None:
                self.bucket_hook(start, end)
        else:
            work = tmp = None
            if self.collective:
                tmp = view.to(self.comm_dtype) if self.comm_dtype is not None else None
                work = dist.all_reduce(view if tmp is None else tmp, group=self.group, async_op=True)
            self._pending.append((work, tmp, start, end))

    def _all_reduce(self, view: torch.Tensor) -> None:
        if self.comm_dtype is not None:
            tmp = view.to(self.comm_dtype)
            coll.all_reduce(tmp, group=self.group)
            view.copy_(tmp)
        else:
            coll.all_reduce(view, group=self.group)

    def finish(self) -> None:
        """Launch any remaining bucket and make the compute stream wait for all of them."""
        if not self.enabled:
            return
        self.mark_ready(self.flat.numel())
        if self.is_cuda and self.overlap and self._comm_stream is not None:
            cur = torch.cuda.current_stream(self.flat.device)
            pair = None
            if self._timing is not None:
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                pair[0].record(cur)
            for idx in self._pending:
                cur.wait_event(self._done_events[idx])
            if pair is not None:
                pair[1].record(cur)
                self._timing.append(pair)
        elif not self.is_cuda:
            for work, tmp, start, end in self._pending:
                if work is not None:
                    work.wait()
                if tmp is not None:
                    self.flat[start:end].copy_(tmp)
                if self.bucket_hook is not None:
                    self.bucket_hook(start, end)
        self._pending.clear()

    # ---------------------------------------------------------------------------------
    def start_timing(self) -> None:
        """From now on :meth:`finish` brackets its waits with a timed event pair: the time the
        compute stream stands still for communication (and the per-bucket hooks) that backward
        did not hide."""
        self._timing = []

    def exposed_ms(self) -> Optional[float]:
        """Mean exposed wait per step since :meth:`start_timing` (synchronises); None if nothing
        was timed."""
        if not self._timing:
            return None
        torch.cuda.synchronize(self.flat.device)
        total = sum(a.elapsed_time(b) for a, b in self._timing)
        return total / len(self._timing)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world
