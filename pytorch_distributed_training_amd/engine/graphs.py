"""HIP-graph capture of a training step as a *segmented program*.

The native engine's step is a fixed sequence of kernel launches for a given batch shape, so
its host cost (hundreds of Python -> ctypes -> hipLaunchKernel round trips) can be paid once:
the step is captured into HIP graphs and replayed.  Collectives stay outside the graphs --
RCCL calls are issued from the host between graph launches -- so a step becomes

    graph 0, host action 0, graph 1, host action 1, ..., graph n

where a host action is a closure over static tensors (a SyncBN all-reduce, the launch of a
gradient bucket on the communication stream, the final wait).  With one rank the whole
forward+backward is a single graph.

Memory: every graph of a program allocates from one private pool; they are replayed in
capture order and never concurrently, which is the condition under which sharing is safe.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch


class StepProgram:
    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.items: List[Tuple[str, object]] = []
        self.pool = torch.cuda.graph_pool_handle()
        self._cur: Optional[torch.cuda.CUDAGraph] = None

    # ------------------------------------------------------------------ capture side
    def begin(self) -> None:
        g = torch.cuda.CUDAGraph()
        # thread-local error mode: the process group's watchdog thread polls events while we
        # capture, which the default (global) mode would treat as a capture violation
        g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self._cur = g

    def _end(self) -> None:
        if self._cur is not None:
            self._cur.capture_end()
            self.items.append(("graph", self._cur))
            self._cur = None

    def host(self, fn: Callable[[], None]) -> None:
        """Close the current graph, run ``fn`` now (un-captured) and record it, open the next."""
        self._end()
        fn()
        self.items.append(("host", fn))
        self.begin()

    def abort(self) -> None:
        if self._cur is not None:
            try:
                self._cur.capture_end()
            except Exception:       # noqa: BLE001 -- the capture is already invalid
                pass
            self._cur = None

    def finish(self) -> None:
        self._end()

    # ------------------------------------------------------------------ replay side
    def replay(self) -> None:
        for kind, item in self.items:
            if kind == "graph":
                item.replay()
            else:
                item()

    @property
    def num_graphs(self) -> int:
        return sum(1 for k, _ in self.items if k == "graph")

    @property
    def num_host_actions(self) -> int:
        return sum(1 for k, _ in self.items if k == "host")


class ListProgram:
    """The step as a flat command list: every library call the HIP backend made while the step
    ran once (kernel launches on their streams, cross-stream waits) plus the host actions, in
    order.  Replaying it is a loop of C calls -- ordinary stream launches, so kernels on the
    second stream overlap as they do eagerly and collectives cost what they cost eagerly, at a
    fraction of the host time of the Python that built the list.

    The recorded step allocates from ``pool`` (a private ``torch.cuda.MemPool``): the addresses in
    the list stay reserved for it, and re-running the list in the same order reuses them the
    same way the recorded step did.
    """

    def __init__(self, device: torch.device, hip_ops) -> None:
        self.device = device
        self._ops = hip_ops
        self.items: List[Tuple[Optional[Callable], object]] = []    # (library fn, args) or (None, host action)
        self.pool = torch.cuda.MemPool()
        self.stream = torch.cuda.current_stream(device).cuda_stream
        self.num_host_actions = 0

    def begin(self) -> None:
        self._ops.begin_recording(self.items)

    def host(self, fn: Callable[[], None]) -> None:
        """A host action (collective, event/stream call, per-bucket optimizer update): run now with
        recording suspended -- library calls it makes belong to the action, which replays them
        itself -- and listed as one item whose return value means nothing."""
        self._ops.end_recording()
        try:
            fn()
        finally:
            self._ops.begin_recording(self.items)
        self.items.append((None, fn))
        self.num_host_actions += 1

    def finish(self) -> None:
        self._ops.end_recording()

    abort = finish

    def replay(self) -> None:
        if torch.cuda.current_stream(self.device).cuda_stream != self.stream:
            raise RuntimeError("a recorded step must be replayed on the stream it was recorded on")
        for fn, args in self.items:
            if fn is None:
                args()                     # host action: whatever it returns (e.g. Work.wait() -> True) is not a status
            elif fn(*args):                # library call: non-zero is a hipError_t / argument error
                raise RuntimeError(f"replayed call {getattr(fn, '__name__', fn)} failed")

    @property
    def num_graphs(self) -> int:
        return 0

    @property
    def num_calls(self) -> int:
        return len(self.items) - self.num_host_actions
