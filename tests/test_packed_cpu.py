"""Packed image store and the GPU-side augmentation's host half (box sampling, reference resize, prefetcher
transform, trainer wiring) on CPU.  The HIP kernel itself is checked against ``ops.ref.crop_resize_flip`` in
``tests/test_hip_ops.py``."""
import os

import numpy as np
import pytest
import torch

from pytorch_distributed_training_amd.data import (DevicePrefetcher, GpuAugment, PackedImageDataset, build_loader,
                                                   get_dataset, make_iter_dataloader, pack_image_folder)
from pytorch_distributed_training_amd.data.packed import random_resized_crop_boxes
from pytorch_distributed_training_amd.ops import ref


def _image_folder(root, n_per_class=3, sizes=((60, 80), (90, 70), (64, 64))):
    from PIL import Image
    for split in ("train", "val"):
        for ci, cname in enumerate(("n01", "n02")):
            d = root / split / cname
            d.mkdir(parents=True)
            for j in range(n_per_class):
                h, w = sizes[j % len(sizes)]
                arr = np.random.RandomState(ci * 10 + j).randint(0, 255, (h, w, 3), dtype=np.uint8)
                Image.fromarray(arr).save(d / f"img{j}.png")
    return root


def test_pack_and_read_back(tmp_path):
    src = _image_folder(tmp_path / "raw")
    out = tmp_path / "packed"
    for split in ("train", "val"):
        pack_image_folder(str(src), split, str(out), side=48, workers=1)
    tr = get_dataset("packed", str(out), "train", image_size=32, fmt="uint8_nhwc")
    va = get_dataset("packed", str(out), "val", image_size=32, fmt="float_nchw")
    assert isinstance(tr, PackedImageDataset) and len(tr) == 6 and len(va) == 6 and tr.gpu_augment
    rec, label = tr[4]
    assert rec.shape == (48, 48, 3) and rec.dtype == torch.uint8 and label == 1
    # a square source image is stored as its plain resize: compare against PIL directly
    from PIL import Image
    with Image.open(src / "train" / "n01" / "img2.png") as im:
        want = np.asarray(im.convert("RGB").resize((48, 48), Image.BILINEAR))
    assert np.array_equal(tr[2][0].numpy(), want)
    xv, yv = va[0]
    assert xv.shape == (3, 32, 32) and xv.dtype == torch.float32 and yv == 0
    # validation = centre crop of the stored record
    u8 = get_dataset("packed", str(out), "val", image_size=32, fmt="uint8_nhwc")
    rec0 = PackedImageDataset(str(out), "val", 48, "uint8_nhwc")[0][0]
    assert torch.equal(u8[0][0], rec0[8:40, 8:40])
    with pytest.raises(FileNotFoundError):
        get_dataset("packed", str(tmp_path / "nope"), "train", fmt="uint8_nhwc")
    with pytest.raises(ValueError):
        get_dataset("packed", str(out), "train", image_size=64, fmt="uint8_nhwc")       # larger than the records
    with pytest.raises(ValueError):
        get_dataset("packed", str(out), "train", image_size=32, fmt="float_nchw")       # raw records are uint8


def test_crop_boxes_follow_the_random_resized_crop_law():
    rng = np.random.default_rng(0)
    b = random_resized_crop_boxes(20000, 256, rng)
    x, y, w, h, flip = (b[:, i].astype(np.float64) for i in range(5))
    assert (x >= 0).all() and (y >= 0).all() and (w >= 1).all() and (h >= 1).all()
    assert (x + w <= 256).all() and (y + h <= 256).all()
    frac = w * h / 256.0 ** 2
    assert 0.07 < frac.min() and frac.max() <= 1.0 and 0.45 < frac.mean() < 0.62     # uniform(0.08, 1) with rejections
    aspect = w / h
    assert 0.70 < aspect.min() and aspect.max() < 1.40
    assert 0.48 < flip.mean() < 0.52
    again = random_resized_crop_boxes(20000, 256, np.random.default_rng(0))
    assert np.array_equal(b, again)


def test_reference_crop_resize_flip():
    g = torch.Generator().manual_seed(1)
    src = torch.randint(0, 256, (3, 40, 40, 3), generator=g, dtype=torch.uint8)
    whole = torch.tensor([[0, 0, 40, 40, 0], [0, 0, 40, 40, 1], [8, 4, 20, 30, 0]], dtype=torch.int32)
    out = ref.crop_resize_flip(src, whole, 40)
    assert torch.equal(out[0], src[0])                           # identity box at the stored size: exact copy
    assert torch.equal(out[1], src[1].flip(1))
    small = ref.crop_resize_flip(src, whole, 16)
    assert small.shape == (3, 16, 16, 3) and small.dtype == torch.uint8
    # a constant image stays constant under any box
    const = torch.full((1, 40, 40, 3), 77, dtype=torch.uint8)
    assert (ref.crop_resize_flip(const, torch.tensor([[3, 5, 17, 29, 1]], dtype=torch.int32), 24) == 77).all()


def test_gpu_augment_in_the_prefetcher_and_resume(tmp_path):
    src = _image_folder(tmp_path / "raw", n_per_class=4)
    out = tmp_path / "packed"
    pack_image_folder(str(src), "train", str(out), side=48, workers=1)
    ds = get_dataset("packed", str(out), "train", image_size=32, fmt="uint8_nhwc")
    loader = build_loader(ds, 4, 0, True, 1, 0, seed=5, pin_memory=False)
    aug = GpuAugment(32, seed=5, rank=0, ops=ref)
    it = DevicePrefetcher(make_iter_dataloader(loader), torch.device("cpu"), transform=aug)
    batches = [next(it) for _ in range(3)]
    assert all(img.shape == (4, 32, 32, 3) and img.dtype == torch.uint8 for img, _ in batches)
    assert aug.batches >= 3
    # a second stream with the same seed, fast-forwarded, cuts the same crops as the first one did at that position
    loader2 = build_loader(ds, 4, 0, True, 1, 0, seed=5, pin_memory=False)
    aug2 = GpuAugment(32, seed=5, rank=0, ops=ref)
    aug2.skip(2, 4, 48)
    it2 = DevicePrefetcher(make_iter_dataloader(loader2, 0, 2), torch.device("cpu"), transform=aug2)
    img3, lab3 = next(it2)
    assert torch.equal(lab3, batches[2][1]) and torch.equal(img3, batches[2][0])
    # other ranks draw other boxes
    other = GpuAugment(32, seed=5, rank=1, ops=ref)
    raw = torch.stack([ds[i][0] for i in range(4)])
    a, _ = GpuAugment(32, seed=5, rank=0, ops=ref)(raw, torch.zeros(4))
    b, _ = other(raw, torch.zeros(4))
    assert not torch.equal(a, b)
    # batches that are not stored records pass through untouched
    val = torch.zeros(2, 32, 32, 3, dtype=torch.uint8)
    assert aug(val, torch.zeros(2))[0] is val


def test_cpu_augment_variant_uses_worker_streams(tmp_path):
    src = _image_folder(tmp_path / "raw")
    out = tmp_path / "packed"
    pack_image_folder(str(src), "train", str(out), side=48, workers=1)
    ds = get_dataset("packed", str(out), "train", image_size=32, fmt="float_nchw", cpu_augment=True)
    assert not ds.gpu_augment and hasattr(ds.transform, "reseed")
    x, y = ds[0]
    assert x.shape == (3, 32, 32) and x.dtype == torch.float32
    loader = build_loader(ds, 2, 2, True, 1, 0, seed=1, pin_memory=False)
    img, label = next(iter(loader))
    assert img.shape == (2, 3, 32, 32)


def test_trainer_runs_on_a_packed_dataset(tmp_path):
    """The CLI trains and validates from packed records (reference ops on CPU do the crop)."""
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = _image_folder(tmp_path / "raw", n_per_class=8)
    out = tmp_path / "packed"
    for split in ("train", "val"):
        pack_image_folder(str(src), split, str(out), side=40, workers=1)
    cfg = {
        "dataset": {"name": "packed", "root": str(out), "n_classes": 2, "image_size": 32, "input_format": "uint8_nhwc"},
        "training": {"optimizer": {"name": "SGD", "lr": 0.02, "weight_decay": 1e-4, "momentum": 0.9},
                     "lr_schedule": {"name": "multi_step", "milestones": [3], "gamma": 0.1}, "train_iters": 4,
                     "print_interval": 1, "val_interval": 2, "batch_size": 4, "num_workers": 0, "sync_bn": False},
        "validation": {"batch_size": 4, "num_workers": 0},
        "model": {"name": "ResNet18"},
    }
    path = tmp_path / "cfg.yml"
    path.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    env["OMP_NUM_THREADS"] = "2"
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    log_dir = tmp_path / "run"
    res = subprocess.run([sys.executable, os.path.join(root, "train_distributed.py"), "--cfg-filepath", str(path),
                          "--log-dir", str(log_dir), "--file-name-cfg", "t", "--seed", "3", "--dist-backend", "gloo"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    log = open(log_dir / "t.log").read()
    assert "Iter [3/4]" in log and log.count("Acc@1:") >= 1
