"""The streams of one rank, by role, within a fixed budget.

A process on ROCm gets ``GPU_MAX_HW_QUEUES`` hardware queues (4 by default); HIP maps streams onto them, and two streams
that share a queue are served in the order the host submitted their packets.  That is harmless for ordinary kernels, but a
rank also submits RENDEZVOUS kernels -- a kernel that ends only once its peer's matching kernel runs: the in-kernel xGMI
statistics exchange on the compute stream, RCCL's bucket all-reduce on RCCL's own stream.  Behind a spinning rendezvous
kernel, a stream that shares its queue stands still.

Two things keep that safe and cheap:

* **Order (correctness, any number of queues).**  One host thread per rank submits all GPU work, and every rank runs the
  same program, so rendezvous operations R1, R2, ... are submitted in the SAME order on all ranks.  By induction R_i
  starts everywhere: whatever sits in front of it in its queue was submitted earlier, i.e. is ordinary work (finite) or a
  rendezvous R_j, j < i (completed everywhere by the induction hypothesis); cross-stream waits only name events recorded
  earlier in host order.  A rendezvous kernel never occupies the chip (one workgroup for the exchange, a few for RCCL), so a
  started R_i on one rank cannot keep R_i on another rank from being dispatched.  No cycle of waits can form, whatever the
  stream-to-queue mapping is -- the shared-GPU rehearsal in tests/test_stream_budget_gpu.py runs with TWO queues.
  What WOULD break this is a second host thread submitting GPU work (its packets would interleave differently on each
  rank): loaders, checkpoint writers and loggers in this package never touch the GPU from a side thread.
* **Budget (performance).**  A rank owns exactly four streams: compute (the current stream), ``wgrad`` (weight gradients
  beside the critical path), ``comm`` (bucket events, the per-bucket optimizer) and ``copy`` (the next batch's upload); a
  fifth role is refused.  In a multi-rank RCCL job RCCL's internal stream comes on top, i.e. five streams on four hardware
  queues, so two of them share a queue.  Folding the copy into ``comm`` or ``wgrad`` to stay at four was measured and is
  WORSE: with every collective really issued through RCCL the step takes 62.0 ms (copy on ``comm``) or 64.0 ms (on
  ``wgrad``) against 58.5 ms with a copy stream of its own (plain: 57.7; profiles/ab_copy_stream_r3.txt) -- an upload
  queued on a stream that also carries kernels waits behind them and is waited for by the next step.  A shared hardware
  queue costs only what one stream's packets delay the other's, and by the order argument above it cannot deadlock.
  (``PDT_COPY_STREAM=comm|wgrad`` reproduces the measurement.)

Upstream has no counterpart: it runs everything on the default stream plus DDP's internal ones
(reference ``train_distributed.py:198,253-254``).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

BUDGET = 4
_SLOT = {"capture": "wgrad"}          # roles that never coexist and therefore share one slot of the budget

_streams: Dict[Tuple[int, str], "torch.cuda.Stream"] = {}


def _index(device: torch.device) -> int:
    return device.index if device.index is not None else torch.cuda.current_device()


def _library_streams() -> int:
    """Streams the collective library runs on besides ours (RCCL: one per device and communicator)."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl" \
            and (dist.get_world_size() > 1 or os.environ.get("PDT_FORCE_COLLECTIVES") == "1"):
        return 1
    return 0


def in_use(device: torch.device) -> Dict[str, int]:
    """``{role: stream id}`` of the streams handed out on ``device`` (plus ``compute`` and, if any, ``rccl``)."""
    dev = _index(torch.device(device))
    out = {"compute": 0}
    out.update({role: id(s) for (d, role), s in _streams.items() if d == dev})
    if _library_streams():
        out["rccl"] = -1
    return out


def count(device: torch.device) -> int:
    dev = _index(torch.device(device))
    own = {id(s) for (d, _), s in _streams.items() if d == dev}
    return 1 + len(own) + _library_streams()


def get(role: str, device: torch.device, priority: Optional[int] = None) -> "torch.cuda.Stream":
    """The stream for ``role`` on ``device`` (created on first use, the same object afterwards).  Roles: ``wgrad``,
    ``comm``, ``copy``, ``capture`` (HIP-graph capture; shares the slot of ``wgrad``, which graph replay does not use)."""
    device = torch.device(device)
    dev = _index(device)
    slot = _SLOT.get(role, role)
    s = _streams.get((dev, slot))
    if s is not None:
        return s
    mode = os.environ.get("PDT_COPY_STREAM")        # A/B measurements: the copy on "comm" / "wgrad" instead of its own
    if role == "copy" and mode in ("comm", "wgrad"):
        s = get(mode, device)
    else:
        if count(device) - _library_streams() >= BUDGET:
            raise RuntimeError(f"stream budget exceeded: {sorted(in_use(device))} are in use on cuda:{dev}, no room for "
                               f"{role!r} (at most {BUDGET} streams per rank besides the collective library's)")
        s = _masked_stream(role, device) or \
            torch.cuda.Stream(device=device, **({"priority": int(priority)} if priority is not None else {}))
    _streams[(dev, slot)] = s
    return s


_keep_alive = []


def _masked_stream(role: str, device: torch.device):
    """Measurement aid: ``PDT_CU_MASK_<ROLE>=<hex words, comma separated, least significant first>`` creates the role's
    stream on a subset of the compute units (``hipExtStreamCreateWithCUMask``).  None when not asked for."""
    spec = os.environ.get(f"PDT_CU_MASK_{role.upper()}")
    if not spec:
        return None
    import ctypes
    words = [int(w, 16) for w in spec.split(",")]
    arr = (ctypes.c_uint32 * len(words))(*words)
    hip = ctypes.CDLL("libamdhip64.so")
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with status {rc}")
    _keep_alive.append((hip, arr))
    return torch.cuda.ExternalStream(handle.value, device=device)


def reset() -> None:
    """Forget every stream (tests)."""
    _streams.clear()
