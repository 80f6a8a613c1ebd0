"""Process launcher.

Upstream's ``Runner.__call__`` spawns one worker per visible GPU with
``mp.spawn(self.worker, nprocs=ngpus_per_node, start_method="spawn")`` when the job is
distributed and otherwise runs ``worker(0)`` in-process; a failing child makes ``spawn``
terminate its siblings and re-raise in the parent (reference ``train_distributed.py:125-135``).

Here the launcher supports both that flag-driven mode and torchrun-style environment
launch (``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*``), in which every process is already
a worker.  Failure semantics are fail-fast: any worker exception ends the whole job with
a non-zero exit status; log files are kept.
"""
from __future__ import annotations

import os
from functools import partial
from typing import Any, Dict, Optional

import torch
import torch.multiprocessing as mp

from .logging_utils import get_tb_writer
from .parallel import dist_env

START_METHOD = "spawn"


def resolve_topology(num_nodes: int, node_rank: int, multiprocessing: bool, nproc_per_node: Optional[int],
                     backend: str) -> Dict[str, int]:
    """World layout from the upstream-style flags (``-1`` defaults mean single node, rank 0)."""
    num_nodes = num_nodes if num_nodes and num_nodes > 0 else 1
    node_rank = node_rank if node_rank and node_rank >= 0 else 0
    if multiprocessing:
        if nproc_per_node and nproc_per_node > 0:
            ppn = nproc_per_node
        else:
            ppn = torch.cuda.device_count() if backend != "gloo" else 1
            ppn = max(ppn, 1)
    else:
        ppn = 1
    return {"num_nodes": num_nodes, "node_rank": node_rank, "procs_per_node": ppn, "world_size": num_nodes * ppn}


def _spawned_worker(local_rank: int, cfg: Dict[str, Any], topo: Dict[str, int], backend: str, dist_url: str,
                    seed: Optional[int], log_dir: Optional[str], file_name_cfg: Optional[str], logger_queue,
                    ops_backend: str) -> None:
    from .trainer import run_worker

    rank = dist_env.global_rank(topo["node_rank"], topo["procs_per_node"], local_rank)
    info = dist_env.init_distributed(backend, dist_url, topo["world_size"], rank, local_rank,
                                     topo["procs_per_node"], cfg["runtime"].get("collective_timeout_s", 1800))
    try:
        run_worker(cfg, info, logger_queue, partial(get_tb_writer, log_dir, file_name_cfg), seed, log_dir,
                   ops_backend)
    finally:
        if info.distributed:
            import torch.distributed as dist

            if dist.is_initialized():
                dist.destroy_process_group()


def launch(cfg: Dict[str, Any], num_nodes: int, node_rank: int, dist_url: str, backend: str, multiprocessing: bool,
           seed: Optional[int], log_dir: Optional[str], file_name_cfg: Optional[str], logger_queue,
           nproc_per_node: Optional[int] = None, ops_backend: str = "auto", logger=None) -> None:
    if dist_env.env_launch_detected():
        # torchrun: this process IS a worker.
        from .trainer import run_worker

        info = dist_env.init_from_env(backend, cfg["runtime"].get("collective_timeout_s", 1800))
        if logger:
            logger.info("Start from environment launch: rank %d / %d", info.rank, info.world_size)
        try:
            run_worker(cfg, info, logger_queue, partial(get_tb_writer, log_dir, file_name_cfg), seed, log_dir,
                       ops_backend)
        finally:
            dist_env.shutdown() if info.distributed else None
        return

    topo = resolve_topology(num_nodes, node_rank, multiprocessing, nproc_per_node, backend)
    args = (cfg, topo, backend, dist_url, seed, log_dir, file_name_cfg, logger_queue, ops_backend)
    if topo["world_size"] > 1 and topo["procs_per_node"] > 1:
        if logger:
            logger.info("Start from multiprocessing: %d processes on this node, world size %d",
                        topo["procs_per_node"], topo["world_size"])
        mp.spawn(_spawned_worker, args=args, nprocs=topo["procs_per_node"], start_method=START_METHOD)
    else:
        if logger:
            logger.info("Start from direct call (world size %d)", topo["world_size"])
        _spawned_worker(0, *args)
