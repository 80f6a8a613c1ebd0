"""Process topology, rendezvous and small collective helpers.

Upstream derives the world size as ``num_nodes * gpus_per_node`` under
``--multiprocessing``, the global rank as ``node_rank * gpus_per_node + gpu_id`` and joins
with ``dist.init_process_group(backend, init_method=tcp://..., world_size, rank)``
(reference ``train_distributed.py:116-122,143-154``).  Its one-process-per-node mode is
broken (every spawned child gets the same rank), so here the world is always
``nodes x processes_per_node``.

Two launch styles are supported: the upstream flags (``--num-nodes --rank --dist-url
--multiprocessing``, processes spawned by :mod:`..launch`) and torchrun-style environment
variables (``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT``).  The backend
``"nccl"`` is RCCL on ROCm builds; ``"gloo"`` gives the CPU path.
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass
class DistInfo:
    rank: int = 0
    local_rank: int = 0
    world_size: int = 1
    procs_per_node: int = 1
    backend: str = "nccl"
    device: torch.device = torch.device("cpu")

    @property
    def distributed(self) -> bool:
        return self.world_size > 1

    @property
    def is_main(self) -> bool:
        return self.rank == 0


def env_launch_detected() -> bool:
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def global_rank(node_rank: int, procs_per_node: int, local_rank: int) -> int:
    return node_rank * procs_per_node + local_rank


def pick_device(local_rank: int, backend: str, device_type: Optional[str] = None) -> torch.device:
    """``device_type="cuda"`` puts the rank on a GPU whatever the backend (gloo reduces GPU tensors
    through the host: several ranks can then share one GPU, which is how the multi-rank paths are
    exercised on a one-GPU machine); by default gloo means the CPU path."""
    want_gpu = device_type == "cuda" or (device_type is None and backend != "gloo")
    if want_gpu and torch.cuda.device_count() > 0 and torch.cuda.is_available():
        dev = torch.device("cuda", local_rank % torch.cuda.device_count())
        torch.cuda.set_device(dev)
        return dev
    return torch.device("cpu")


def init_distributed(backend: str, dist_url: Optional[str], world_size: int, rank: int, local_rank: int,
                     procs_per_node: int, timeout_s: int = 1800, device_type: Optional[str] = None) -> DistInfo:
    """Select the device, then join the process group when ``world_size > 1``."""
    if backend == "nccl" and not (torch.cuda.device_count() > 0 and torch.cuda.is_available()):
        backend = "gloo"          # no GPU visible: CPU plumbing path
    device = pick_device(local_rank, backend, device_type)
    forced = os.environ.get("PDT_FORCE_COLLECTIVES") == "1"      # debug: one-rank group, see GradReducer
    if (world_size > 1 or forced) and not dist.is_initialized():
        kwargs = dict(backend=backend, world_size=world_size, rank=rank,
                      timeout=datetime.timedelta(seconds=timeout_s))
        if dist_url and dist_url != "env://":
            kwargs["init_method"] = dist_url
        else:
            kwargs["init_method"] = "env://"
            if world_size == 1:          # forced one-rank group: no launcher has set the rendezvous
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            try:
                dist.init_process_group(device_id=device, **kwargs)
            except TypeError:  # older signature without device_id
                dist.init_process_group(**kwargs)
        else:
            dist.init_process_group(**kwargs)
    return DistInfo(rank=rank, local_rank=local_rank, world_size=world_size, procs_per_node=procs_per_node,
                    backend=backend, device=device)


def init_from_env(backend: str = "nccl", timeout_s: int = 1800, device_type: Optional[str] = None) -> DistInfo:
    """torchrun-style initialisation from ``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*``."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(max(torch.cuda.device_count(), 1))))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    return init_distributed(backend, "env://", world, rank, local_rank, local_world, timeout_s, device_type)


def shutdown() -> None:
    if dist.is_available() and dist.is_initialized():
        try:
            dist.barrier()
        except Exception:
            pass
        dist.destroy_process_group()


def tensor_fingerprint(t: torch.Tensor) -> torch.Tensor:
    """Cheap order-sensitive checksum of a flat fp32 buffer (two fp64 numbers)."""
    x = t.detach().double()
    w = torch.arange(1, x.numel() + 1, device=x.device, dtype=torch.float64)
    return torch.stack([x.sum(), (x * torch.remainder(w, 9973.0)).sum()])


def check_replicas_in_sync(t: torch.Tensor, group=None, rtol: float = 0.0) -> bool:
    """Debug check: do all ranks hold the same buffer?  (min == max of the fingerprints.)

    Stands in for the consistency DDP gives implicitly by broadcasting parameters at
    construction and buffers before every forward
    (``torch/nn/parallel/distributed.py:864-871,2178-2221``).
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    fp = tensor_fingerprint(t)
    lo, hi = fp.clone(), fp.clone()
    from . import collectives as coll
    coll.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    coll.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    tol = rtol * hi.abs().max().item()
    return bool(((hi - lo).abs() <= tol).all().item())
