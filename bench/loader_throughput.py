"""Input-pipeline throughput: can the loader feed one MI355X (>13 000 images/s on ResNet-50)?

    python bench/loader_throughput.py [--images 1024] [--workers 16] [--batch 256] [--seconds 6]

Builds a synthetic JPEG ImageFolder (ImageNet-like 500 x 375 pictures with smooth content, so that the files
have realistic sizes and decode costs), then measures images/s of

  jpeg+pil      the upstream-style path: JPEG decode + RandomResizedCrop + flip in ``--workers`` loader processes;
  packed+cpu    packed fixed-size records, crop/resize with PIL in the workers (no decode);
  packed+gpu    packed records copied as they are; crop/resize/flip in one HIP kernel behind the upload (the host
                does a memcpy per image);
  packed+gpu (main process only)   the same with ``num_workers = 0``: what ONE core sustains.

The GPU rows include pinned-memory staging, the host-to-device copy and the kernel; with no GPU present they run
the reference crop on the CPU and say so.
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_jpeg_folder(root: str, n: int, classes: int = 8) -> None:
    from PIL import Image, ImageFilter
    rs = np.random.RandomState(0)
    for split, count in (("train", n), ("val", max(classes, n // 8))):
        for i in range(count):
            d = os.path.join(root, split, f"n{i % classes:03d}")
            os.makedirs(d, exist_ok=True)
            w, h = (500, 375) if i % 3 else (375, 500)
            base = rs.randint(0, 255, (h // 8, w // 8, 3), dtype=np.uint8)
            im = Image.fromarray(base).resize((w, h), Image.BICUBIC).filter(ImageFilter.GaussianBlur(1))
            noise = rs.randint(-12, 12, (h, w, 3))
            im = Image.fromarray(np.clip(np.asarray(im).astype(np.int16) + noise, 0, 255).astype(np.uint8))
            im.save(os.path.join(d, f"img{i:06d}.JPEG"), quality=90)


def rate(it, seconds: float, batch: int, sync=None) -> float:
    for _ in range(2):
        next(it)
    if sync:
        sync()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        next(it)
        n += batch
    if sync:
        sync()
    return n / (time.perf_counter() - t0)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=1024)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--side", type=int, default=256)
    args = ap.parse_args()
    from pytorch_distributed_training_amd.data import (DevicePrefetcher, GpuAugment, build_loader, get_dataset,
                                                       make_iter_dataloader, pack_image_folder)

    gpu = torch.cuda.is_available()
    dev = torch.device("cuda", 0) if gpu else torch.device("cpu")
    sync = torch.cuda.synchronize if gpu else None
    tmp = tempfile.mkdtemp(prefix="pdt_loader_")
    raw, packed = os.path.join(tmp, "raw"), os.path.join(tmp, "packed")
    t0 = time.perf_counter()
    make_jpeg_folder(raw, args.images)
    t1 = time.perf_counter()
    pack_image_folder(raw, "train", packed, side=args.side, workers=min(args.workers, 16))
    t2 = time.perf_counter()
    size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(os.path.join(raw, "train")) for f in fs)
    print(f"{args.images} JPEGs ({size / args.images / 1e3:.0f} KB each) written in {t1 - t0:.1f} s, packed to "
          f"{args.side}x{args.side} records in {t2 - t1:.1f} s ({args.images / (t2 - t1):.0f} images/s with "
          f"{min(args.workers, 16)} processes)")
    print(f"device: {dev}; loader workers: {args.workers}; batch {args.batch}; CPUs visible: {os.cpu_count()}")
    rows = []

    def run(label, ds, workers, transform=None):
        loader = build_loader(ds, args.batch, workers, True, 1, 0, seed=1, pin_memory=gpu)
        it = DevicePrefetcher(make_iter_dataloader(loader), dev, transform=transform)
        r = rate(it, args.seconds, args.batch, sync)
        rows.append((label, workers, r))
        print(f"{label:34s} workers {workers:3d}   {r:9.0f} images/s   ({r / max(workers, 1):7.0f} per process)", flush=True)
        del it, loader

    run("jpeg+pil", get_dataset("imagenet", raw, "train", image_size=224, fmt="uint8_nhwc"), args.workers)
    run("packed+cpu", get_dataset("packed", packed, "train", image_size=224, fmt="uint8_nhwc", cpu_augment=True),
        args.workers)
    ds = get_dataset("packed", packed, "train", image_size=224, fmt="uint8_nhwc")
    run("packed+gpu" if gpu else "packed+reference crop on CPU", ds, args.workers, GpuAugment(224, 1, 0))
    run("packed+gpu (main process only)" if gpu else "packed+reference crop (main only)", ds, 0, GpuAugment(224, 1, 0))
    need = 13500.0
    print(f"\nto feed {need:.0f} images/s (one MI355X, ResNet-50 bf16):")
    for label, workers, r in rows:
        per = r / max(workers, 1)
        print(f"  {label:34s} {need / per:7.1f} processes at the measured per-process rate")
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
