"""Low-latency all-reduce of small fp32 vectors between the GPUs of one node.

SyncBatchNorm issues two collectives of at most 16 KB per layer, each on the critical path
(upstream: ``SyncBatchNorm`` forward all-gather and backward all-reduce,
``torch/nn/modules/_functions.py:65-79,157-165`` behind reference ``train_distributed.py:196-197``).
Over RCCL each costs a launch on the communication library's terms plus a ring/tree
protocol; over xGMI peer mappings it is one kernel on the compute stream (see
``ops/csrc/xgmi_allreduce.hip``), which also makes it capturable in the step's HIP graph.

The path is only used after it has proved itself on the machine at hand: :meth:`self_test`
compares it against the process group's own all-reduce on every rank and the ranks agree
on the verdict; any failure (IPC not available, ranks on different hosts, a mismatch, a
timeout) leaves the caller on the library collective.
"""
from __future__ import annotations

import ctypes
import os
import socket
from typing import Optional

import torch
import torch.distributed as dist

MAX_WORLD = 8
IPC_HANDLE_BYTES = 64


def _agree(value: int, op, group, device: torch.device) -> int:
    """All-reduce one control integer (on the host for gloo groups, on the GPU for RCCL)."""
    ctrl = torch.device("cpu") if dist.get_backend(group) == "gloo" else device
    t = torch.tensor([value], dtype=torch.int32, device=ctrl)
    dist.all_reduce(t, op=op, group=group)
    return int(t.item())


class XgmiAllReduce:
    def __init__(self, lib, handle: int, world: int, rank: int, max_floats: int, device: torch.device, group) -> None:
        self._lib = lib
        self._h = ctypes.c_void_p(handle)
        self.world, self.rank, self.max_floats = world, rank, max_floats
        self.device = device
        self.group = group
        self.calls = 0

    # ------------------------------------------------------------------ construction
    @classmethod
    def create(cls, group, device: torch.device, max_floats: int = 4096) -> Optional["XgmiAllReduce"]:
        """Collective over ``group``: build the peer mappings, or return None on every rank."""
        if device.type != "cuda" or not dist.is_initialized():
            return None
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        try:
            from ..ops import hip as hip_ops
        except ImportError:
            hip_ops = None
        lib = getattr(hip_ops, "_lib", None)
        handle = ctypes.c_void_p()
        ok = lib is not None and world <= MAX_WORLD
        if ok:
            torch.cuda.set_device(device)
            ok = lib.pdt_car_create(world, rank, max_floats, ctypes.byref(handle)) == 0
        mine = ctypes.create_string_buffer(IPC_HANDLE_BYTES)
        if ok and world > 1:
            ok = lib.pdt_car_ipc_handle(handle, mine) == 0
        # one exchange carries the verdict so far, the host name and the handle
        info = [None] * world
        my_dev = device.index if device.index is not None else torch.cuda.current_device()
        dist.all_gather_object(info, (bool(ok), socket.gethostname(), bytes(mine.raw), my_dev), group=group)
        ok = all(i[0] for i in info) and len({i[1] for i in info}) == 1
        # Ranks that SHARE a GPU (a one-GPU rehearsal of a multi-rank job) are refused: a kernel
        # spinning on a peer's flag depends on the other process's kernels being scheduled next
        # to it, and with host-blocking collectives (gloo) in the same step that arrangement was
        # seen to stall for tens of seconds.  The protocol tests opt in explicitly.
        if ok and len({i[3] for i in info}) < world and os.environ.get("PDT_XGMI_ALLOW_SHARED_GPU") != "1":
            ok = False
        if ok:       # every peer GPU must be directly addressable from this one (xGMI / PCIe P2P)
            for i in info:
                if i[3] != my_dev and not torch.cuda.can_device_access_peer(my_dev, i[3]):
                    ok = False
        if ok and world > 1:
            blob = b"".join(i[2] for i in info)
            ok = lib.pdt_car_open(handle, ctypes.c_char_p(blob)) == 0
        if _agree(1 if ok else 0, dist.ReduceOp.MIN, group, device) != 1:
            if handle.value:
                lib.pdt_car_destroy(handle)
            return None
        return cls(lib, handle.value, world, rank, max_floats, device, group)

    # ------------------------------------------------------------------ use
    def all_reduce(self, t: torch.Tensor) -> None:
        """In-place SUM over the ranks (fp32, contiguous, numel a multiple of 4, at most
        ``max_floats``), enqueued on the current stream."""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("xgmi all_reduce: expected a contiguous fp32 GPU tensor")
        n = t.numel()
        if n % 4 or n > self.max_floats:
            raise ValueError(f"xgmi all_reduce: {n} elements (need a multiple of 4, at most {self.max_floats})")
        from ..ops.hip import _stream
        rc = self._lib.pdt_car_allreduce(self._h, t.data_ptr(), n, _stream())
        if rc != 0:
            raise RuntimeError(f"pdt_car_allreduce failed with status {rc}")
        self.calls += 1

    def all_reduce_large(self, t: torch.Tensor) -> None:
        """In-place SUM of a contiguous fp32 GPU tensor of ANY length: one kernel on the current stream that walks the
        tensor in pieces of the exchange's capacity (one exchange per piece, one workgroup -- a latency path put to
        work as a rendezvous collective for one-GPU rehearsals, not a bandwidth path)."""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("xgmi all_reduce_large: expected a contiguous fp32 GPU tensor")
        from ..ops.hip import _stream
        rc = self._lib.pdt_car_allreduce_chunks(self._h, t.data_ptr(), t.numel(), _stream())
        if rc != 0:
            raise RuntimeError(f"pdt_car_allreduce_chunks failed with status {rc}")
        self.calls += 1

    def supports(self, t: torch.Tensor) -> bool:
        return t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0 and \
            t.numel() <= self.max_floats and t.data_ptr() % 16 == 0

    def status(self) -> int:
        """Synchronises; non-zero when a wait inside a kernel timed out (results invalid)."""
        out = (ctypes.c_uint * 3)()
        if self._lib.pdt_car_query(self._h, out) != 0:
            return -1
        return int(out[1])

    def rereads(self) -> int:
        """Synchronises; how often a payload had to be read again because its checksum did not match yet (the flag had
        overtaken payload words).  Zero between processes on one GPU; on real links it is the measure of how much the
        fence-free ordering is being leaned on.  -1 if the query failed."""
        out = (ctypes.c_uint * 3)()
        if self._lib.pdt_car_query(self._h, out) != 0:
            return -1
        return int(out[2])

    def peek(self) -> int:
        """The status as last written to pinned host memory: no synchronisation, no GPU call; may
        lag the GPU by the kernels still queued.  Non-zero = a wait timed out (results are NaN)."""
        if self._h is None:
            return 0
        return int(self._lib.pdt_car_peek(self._h))

    def healthy(self) -> bool:
        """Collective: True on every rank iff no rank has seen a timeout."""
        return _agree(self.status(), dist.ReduceOp.MAX, self.group, self.device) == 0

    def self_test(self, rounds: int = 48, production_timeout_s: float = 60.0) -> bool:
        """Collective: compare against the process group's all-reduce over many sizes and
        back-to-back calls (slot reuse, flag reuse); every rank returns the same verdict."""
        # Generous while testing: the first launch on a rank includes loading the code object, and
        # the ranks reach this point at slightly different times (the control all-reduce aligns them).
        self._lib.pdt_car_set_timeout_ms(self._h, 5000)
        _agree(1, dist.ReduceOp.MIN, self.group, self.device)
        gen = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        sizes = [4, 128, 512, 1024, 2048, 4096, self.max_floats]
        ok = True
        pairs = []
        for i in range(rounds):
            n = sizes[i % len(sizes)]
            x = torch.randn(n, generator=gen).to(self.device)
            mine = x.clone()
            self.all_reduce(mine)                  # back to back: no host sync between rounds
            pairs.append((mine, x))
        host = dist.get_backend(self.group) == "gloo"      # gloo reduces on the host
        for mine, x in pairs:
            if host:
                x, mine = x.cpu(), mine.cpu()
            dist.all_reduce(x, group=self.group)
            # same addends, possibly another summation order
            if not torch.allclose(mine, x, rtol=1e-5, atol=1e-5 * self.world):
                ok = False
        # bit-identical on every rank (the sum order is fixed): compare against rank 0's copy
        probe = pairs[-1][0].cpu() if host else pairs[-1][0].clone()
        ref = probe.clone()
        dist.broadcast(ref, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        ok = ok and bool(torch.equal(probe, ref)) and self.status() == 0
        agreed = _agree(1 if ok else 0, dist.ReduceOp.MIN, self.group, self.device) == 1
        # In production a rank may legitimately arrive late (a cold file cache under a loader): the wait
        # is bounded by ``runtime.xgmi_timeout_s`` (a minute by default -- it is a kernel spinning on the
        # GPU, so it is NOT given a library collective's half hour; the long host-side pauses, a
        # checkpoint being written or validation starting, are fenced by barriers on the host).
        ms = os.environ.get("PDT_XGMI_TIMEOUT_MS")
        ms = int(ms) if ms else int(min(max(production_timeout_s, 1.0), 2.0e6) * 1000)
        self._lib.pdt_car_set_timeout_ms(self._h, ms)
        return agreed

    def close(self) -> None:
        if self._h is not None and self._h.value:
            self._lib.pdt_car_destroy(self._h)
            self._h = None
