"""Gradient-bucket size sweep for the xGMI all-reduce overlap.

    python bench/bucket_sweep.py --gpus N --model ResNet152 --batch 128 --caps 4,8,16,25,50,100,200

From a plain shell the script starts its N ranks itself (one per GPU, like ``bench.py --gpus N``: the parent makes no
GPU call and relays rank 0's lines); under a launcher that has set RANK / WORLD_SIZE / MASTER_* (``python -m
torch.distributed.run --nproc-per-node N ... bench/bucket_sweep.py``) it is one of the ranks.  ``--cfg-filepath`` takes
model, SyncBN switch and per-GPU batch from a trainer YAML (``config/ResNet101-large-batch.yml``, ...).

For every bucket cap it builds a fresh engine, runs warm-up + timed steps and prints one JSON line
(images/s, ms/step, number of buckets).  A ring all-reduce over point-to-point xGMI links is bound
per link, so few large buckets win on bandwidth while a small first bucket starts the ring early;
the sweep shows where the exposed tail (the last bucket cannot overlap anything) begins to cost
more than the saved launch/latency overhead.  With one rank it still runs (collectives are
skipped) and reports the bucket layout.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_distributed_training_amd.config import get_cfg, validate_cfg  # noqa: E402
from pytorch_distributed_training_amd.engine.engines import build_engine  # noqa: E402
from pytorch_distributed_training_amd.parallel import dist_env  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ResNet152")
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--caps", default="4,8,16,25,50,100,200", help="bucket caps in MiB")
    ap.add_argument("--first-mb", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sync-bn", type=int, default=1)
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start when no launcher has set the rendezvous variables")
    ap.add_argument("--cfg-filepath", default="", help="trainer YAML: model, sync_bn and the per-GPU batch come from it")
    args = ap.parse_args()
    if args.gpus > 1 and not dist_env.env_launch_detected():
        import importlib.util
        spec = importlib.util.spec_from_file_location("pdt_bench", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)                       # (importing it touches no GPU)
        return bench.self_launch(args.gpus, sys.argv[1:], script=os.path.abspath(__file__))
    if args.cfg_filepath:
        file_cfg = get_cfg(args.cfg_filepath)
        tr = file_cfg["training"]
        world = int(os.environ.get("WORLD_SIZE", "1"))
        args.model = file_cfg["model"]["name"]
        args.sync_bn = int(bool(tr.get("sync_bn", False)))
        args.batch = int(tr.get("per_gpu_batch_size") or max(int(tr["batch_size"]) // max(world, 1), 1))
    info = dist_env.init_from_env("nccl") if dist_env.env_launch_detected() else \
        dist_env.init_distributed("nccl", None, 1, 0, 0, 1)
    dev = info.device
    if dev.type != "cuda":
        print("bucket_sweep.py: no GPU visible", file=sys.stderr)
        return 2
    img = torch.randint(0, 256, (args.batch, 224, 224, 3), device=dev, dtype=torch.uint8)
    label = torch.randint(0, 1000, (args.batch,), device=dev)
    for cap in [float(c) for c in args.caps.split(",")]:
        cfg = validate_cfg({
            "dataset": {"name": "synthetic", "n_classes": 1000, "image_size": 224, "input_format": "uint8_nhwc"},
            "training": {"optimizer": {"name": "SGD", "lr": 0.1, "weight_decay": 1e-4, "momentum": 0.9},
                         "lr_schedule": {"name": "multi_step", "milestones": [10 ** 6], "gamma": 0.1},
                         "train_iters": args.steps + args.warmup, "batch_size": args.batch * info.world_size,
                         "per_gpu_batch_size": args.batch, "sync_bn": bool(args.sync_bn), "precision": "bf16",
                         "bucket_cap_mb": cap, "first_bucket_mb": args.first_mb},
            "model": {"name": args.model}})
        torch.manual_seed(1)
        eng = build_engine(cfg, dev)
        for _ in range(args.warmup):
            eng.train_step(img, label)
        if info.distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.train_step(img, label)
        if info.distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if info.distributed:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        if info.rank == 0:
            print(json.dumps({"model": args.model, "bucket_cap_mb": cap, "n_buckets": len(eng.model.buckets),
                              "n_gpus": info.world_size, "per_gpu_batch": args.batch,
                              "images_per_s": round(args.batch * info.world_size * args.steps / float(dt), 1),
                              "ms_per_step": round(1e3 * float(dt) / args.steps, 3)}), flush=True)
        del eng
        torch.cuda.empty_cache()
    dist_env.shutdown()
    return 0


if __name__ == "__main__":
    sys.exit(main())
