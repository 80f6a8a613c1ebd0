"""Headline benchmark: ResNet-50 bf16 training throughput (images/s, whole job) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

For N > 1 there is one rank per GPU.  Either a launcher starts them, e.g.
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N --steps K --warmup W`` (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* are read from the environment), or -- from a plain shell, with none of those set --
this script starts the N ranks itself: the parent makes no GPU call, runs N fresh
interpreters of this file with the rendezvous variables set (the way upstream's trainer
spawns its workers, reference ``train_distributed.py:125-135``, ``train.sh:6-14``), relays
rank 0's JSON line and exits non-zero if any rank failed.  Collectives go over RCCL.

A step is the full training iteration of the trainer: host-to-device copy of a synthetic
uint8 ImageNet-shaped batch (prefetched on a copy stream), fused normalise/layout kernel,
forward, loss, backward with SyncBatchNorm (N > 1) and bucketed gradient all-reduce
overlapped on a side stream, fused SGD step.  Weak scaling: the per-GPU batch is fixed.
Three set-up steps in which the engine records its step program (``--setup-steps``, reported in the JSON line),
W un-timed warm-up steps, then exactly K timed steps bracketed by barrier +
``torch.cuda.synchronize()``; the time is the MAX over ranks; rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pytorch_distributed_training_amd.config import validate_cfg  # noqa: E402
from pytorch_distributed_training_amd.data.loader import DevicePrefetcher  # noqa: E402
from pytorch_distributed_training_amd.engine.engines import build_engine  # noqa: E402
from pytorch_distributed_training_amd.parallel import collectives, dist_env  # noqa: E402


def synthetic_batches(batch: int, image_size: int, n_classes: int, fmt: str, device: torch.device, mode: str,
                      pool: int = 4, seed: int = 0):
    """Endless generator of (img, label) batches with ImageNet's shapes.

    ``mode='h2d'``: batches live in pinned host memory and are copied every step (what a real
    loader delivers); ``mode='device'``: batches are resident on the GPU.
    """
    g = torch.Generator().manual_seed(seed)
    imgs, labels = [], []
    for _ in range(pool):
        if fmt == "uint8_nhwc":
            img = torch.randint(0, 256, (batch, image_size, image_size, 3), generator=g, dtype=torch.uint8)
        else:
            img = torch.randn(batch, 3, image_size, image_size, generator=g)
        lab = torch.randint(0, n_classes, (batch,), generator=g, dtype=torch.int64)
        if mode == "device":
            img, lab = img.to(device), lab.to(device)
        elif device.type == "cuda":
            img, lab = img.pin_memory(), lab.pin_memory()
        imgs.append(img)
        labels.append(lab)
    i = 0
    while True:
        yield imgs[i % pool], labels[i % pool]
        i += 1


def _free_port() -> int:
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def self_launch(n_ranks: int, argv, timeout_s: float = 3000.0, script: str = None) -> int:
    """Start ``n_ranks`` fresh interpreters of this script, one per GPU (ranks share GPUs round
    robin when fewer are visible).  Called before this process has touched the GPU; it never
    does.  Rank 0's stdout is this process's stdout, the other ranks write to stderr."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL and the xGMI exchange need it
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(_free_port()))
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n_ranks)
    procs = []
    for rank in range(n_ranks):
        renv = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__), *argv], env=renv,
                                      stdout=None if rank == 0 else sys.stderr))
    deadline = time.monotonic() + timeout_s
    codes = [None] * n_ranks
    try:
        while any(c is None for c in codes):
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    codes[i] = pr.poll()
            failed = [c for c in codes if c not in (None, 0)]
            if failed or time.monotonic() > deadline:
                break
            time.sleep(0.05)
    finally:
        # fail fast: one dead rank leaves the others blocked in a collective
        for i, pr in enumerate(procs):
            if codes[i] is None:
                pr.terminate()
        for i, pr in enumerate(procs):
            if codes[i] is None:
                try:
                    codes[i] = pr.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pr.kill()
                    codes[i] = pr.wait()
    bad = [(i, c) for i, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", type=str, default="ResNet50")
    ap.add_argument("--batch", type=int, default=1024,
                    help="per-GPU batch size (sized for 288 GB of HBM: 1024 uses 40 GB, keeps every activation below "
                         "the 2 GB the kernels address with 32-bit offsets, and quarters the number of SyncBN "
                         "exchanges and gradient all-reduces per image against 256)")
    ap.add_argument("--engine", type=str, default="native", choices=("native", "torch"))
    ap.add_argument("--input-format", type=str, default="uint8_nhwc", choices=("uint8_nhwc", "float_nchw"))
    ap.add_argument("--data-mode", type=str, default="h2d", choices=("h2d", "device"))
    ap.add_argument("--sync-bn", type=int, default=1, help="SyncBatchNorm when more than one GPU")
    ap.add_argument("--bucket-cap-mb", type=float, default=25.0)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--backend", type=str, default="nccl")
    ap.add_argument("--setup-steps", type=int, default=3,
                    help="untimed steps before the warm-up in which the engine records its step (see main)")
    args = ap.parse_args()

    hang_dump = float(os.environ.get("PDT_HANG_DUMP_S", "0") or 0)
    if hang_dump > 0:                 # debug aid: every thread's Python stack if the run is still going then
        import faulthandler
        faulthandler.dump_traceback_later(hang_dump, exit=False)

    if args.gpus > 1 and not dist_env.env_launch_detected():
        return self_launch(args.gpus, sys.argv[1:])          # nothing here has touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.gpus > 1 or dist_env.env_launch_detected():
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
            return 2
        info = dist_env.init_from_env(args.backend, device_type="cuda")
    else:
        info = dist_env.init_distributed(args.backend, None, 1, 0, 0, 1, device_type="cuda")
    device = info.device
    if device.type != "cuda":
        print("bench.py: no GPU visible", file=sys.stderr)
        return 2

    cfg = validate_cfg({
        "dataset": {"name": "synthetic", "n_classes": 1000, "image_size": args.image_size,
                    "input_format": args.input_format},
        "training": {
            "optimizer": {"name": "SGD", "lr": 0.1, "weight_decay": 1.0e-4, "momentum": 0.9},
            "lr_schedule": {"name": "multi_step", "milestones": [150000, 300000], "gamma": 0.1},
            "train_iters": args.steps + args.warmup, "batch_size": args.batch * max(args.gpus, 1),
            "per_gpu_batch_size": args.batch, "sync_bn": bool(args.sync_bn), "precision": "bf16",
            "bucket_cap_mb": args.bucket_cap_mb, "engine": args.engine,
        },
        "model": {"name": args.model},
    })
    torch.manual_seed(1029)
    engine = build_engine(cfg, device)
    source = synthetic_batches(args.batch, args.image_size, 1000, args.input_format, device, args.data_mode,
                               seed=info.rank)
    batches = DevicePrefetcher(source, device) if args.data_mode == "h2d" else source

    def step():
        img, label = next(batches)
        return engine.train_step(img, label)

    loss = None
    # Set-up, before the warm-up proper: the engine runs the first step of a batch shape eagerly (it fills its lazily
    # built caches), records the second as a command list and replays that from the third on -- what a tracing compiler
    # would do at build time.  Untimed, counted separately from the W warm-up steps and reported as "setup_steps".
    for _ in range(args.setup_steps):
        loss = step()
    for _ in range(args.warmup):
        loss = step()
    if not engine.check_health():                 # collective; a failed exchange would have poisoned the warm-up
        print("bench.py: communication failure during warm-up", file=sys.stderr)
        return 3
    native = args.engine == "native"
    if native:
        engine.model.reducer.start_timing()
    if info.distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    enqueue = time.perf_counter() - t0          # host time to submit the steps (GPU still running)
    if info.distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if info.distributed:
        collectives.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    exposed = engine.model.reducer.exposed_ms() if native else None
    healthy = engine.check_health()
    final_loss = float(loss) if loss is not None else float("nan")
    global_batch = args.batch * info.world_size
    value = global_batch * args.steps / elapsed
    if info.rank == 0:
        print(json.dumps({
            "metric": f"train images/sec (whole node) {args.model} bf16",
            "value": round(value, 2),
            "unit": "images/s",
            "n_gpus": info.world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": f"synthetic ImageNet-shaped ({args.input_format}, {args.data_mode}), random-init weights",
            "config": {
                "model": args.model, "global_batch": global_batch, "per_gpu_batch": args.batch,
                "setup_steps": args.setup_steps,
                "image_size": args.image_size, "seq_len": None,
                "parallelism": f"dp{info.world_size}" + ("+syncbn" if (args.sync_bn and info.world_size > 1) else ""),
                "engine": args.engine, "optimizer": "SGD momentum 0.9 wd 1e-4", "final_loss": round(final_loss, 4),
                "peak_mem_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
                "rccl_world": dist.get_world_size() if dist.is_initialized() else 1,
                "backend": dist.get_backend() if dist.is_initialized() else None,
                "grad_buckets": len(engine.model.buckets) if native else None,
                "grad_buckets_reduced_per_step": (
                    engine.model.reducer.stats["launched"] // max(engine.model.reducer.stats["steps"], 1)
                    if native else None),
                "grad_allreduce_mb_per_step": (
                    round(engine.model.reducer.stats["bytes"] / max(engine.model.reducer.stats["steps"], 1) / 2 ** 20, 2)
                    if native else None),
                # time per step the compute stream stood still at the end of backward waiting for the
                # communication stream (last all-reduce + the per-bucket optimizer updates not yet done)
                "exposed_comm_ms_per_step": round(exposed, 4) if exposed is not None else None,
                "optimizer_in_backward": engine.model.optimizer_in_backward if native else None,
                "comm_healthy": bool(healthy),
                "host_submit_ms_per_step": round(1e3 * enqueue / args.steps, 3),
                "syncbn_comm": engine.model.syncbn_comm if (args.engine == "native" and engine.model.sync_bn) else None,
                "step_replay": engine.model.active_replay_mode() if native else None,
                "sync_bn_active": bool(engine.model.sync_bn) if args.engine == "native" else None,
            },
        }), flush=True)
    if dist.is_initialized():
        dist_env.shutdown()
    return 0 if healthy else 3


if __name__ == "__main__":
    sys.exit(main())
