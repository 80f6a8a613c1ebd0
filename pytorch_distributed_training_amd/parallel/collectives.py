"""Collectives that also work for GPU tensors on a gloo process group.

RCCL ("nccl") reduces GPU tensors on the GPU and is what a real job uses.  gloo is the CPU
backend; it is also how several ranks are rehearsed on ONE GPU (``bench.py --gpus 2 --backend
gloo``, the one-GPU distributed tests).  gloo's own handling of GPU tensors was seen to stall
with two processes on one device, so here a GPU tensor on a gloo group is staged through the
host explicitly: a blocking device-to-host copy on the current stream (it waits for that
stream's earlier work only), the collective on the host copy, a copy back.  Every rank blocks in
the same program order, and no rank's GPU work depends on a peer's, so this cannot deadlock.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


_rendezvous = None        # (XgmiAllReduce, group) of a one-GPU rehearsal: fp32 sums on the GPU go through its kernel


def set_device_rendezvous(xgmi, group=None) -> None:
    """From now on fp32 SUM all-reduces of GPU tensors over ``group`` run as the exchange kernel of ``xgmi`` on the
    current stream (stream-ordered, never blocking the host) -- the stand-in for RCCL when several ranks share one GPU."""
    global _rendezvous
    _rendezvous = (xgmi, group) if xgmi is not None else None


def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce(t: torch.Tensor, op=dist.ReduceOp.SUM, group=None) -> None:
    """In-place all-reduce of ``t`` (blocking for staged tensors, stream-ordered over RCCL)."""
    if _rendezvous is not None and _rendezvous[1] is group and t.is_cuda and t.dtype == torch.float32 \
            and op == dist.ReduceOp.SUM and t.is_contiguous():
        _rendezvous[0].all_reduce_large(t)
        return
    if _staged(t, group):
        host = t.detach().to("cpu")
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)


def broadcast(t: torch.Tensor, src: int = 0, group=None) -> None:
    if _staged(t, group):
        host = t.detach().to("cpu")
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src, group=group)


def is_host_staged(t: torch.Tensor, group=None) -> bool:
    """Would :func:`all_reduce` block the host for this tensor?  (Callers that want an asynchronous
    collective fall back to the blocking form then.)"""
    if _rendezvous is not None and _rendezvous[1] is group and t.is_cuda and t.dtype == torch.float32:
        return False
    return dist.is_available() and dist.is_initialized() and _staged(t, group)
