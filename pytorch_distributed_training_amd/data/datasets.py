"""Datasets: ImageNet folder reader and a synthetic ImageNet-shaped generator.

Upstream builds ``get_dataset(name, root, split)`` for ``split in {"train", "val"}`` and its
loaders yield ``(img, label)`` batches of float NCHW images and integer labels
(reference ``train_distributed.py:171-181,252,309``; ``config/ResNet50.yml:1-4`` names
``imagenet`` under ``~/datasets/ILSVRC2012``).  The usual ImageNet pipeline is assumed:
train = RandomResizedCrop(224) + horizontal flip, val = Resize(256) + CenterCrop(224), both
normalised with the ImageNet mean / std.

torchvision is not a dependency: the folder scan and the transforms are implemented here
on PIL + numpy.  ``name="synthetic"`` generates deterministic pseudo-random images of the
same shape (for benchmarks, smoke runs and tests; no files are read).

Two output formats: ``float_nchw`` (upstream contract: normalised fp32 ``[3, H, W]``) and
``uint8_nhwc`` (raw ``[H, W, 3]`` bytes; normalisation, cast and layout change then happen in
one fused GPU kernel, and the host-to-device copy is 4x smaller).
"""
from __future__ import annotations

import math
import os
import random
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_IMG_EXT = (".jpeg", ".jpg", ".png", ".bmp", ".webp", ".ppm", ".tif", ".tiff")


# ----------------------------------------------------------------------------- transforms
def _random_resized_crop_box(w: int, h: int, rng: random.Random, scale=(0.08, 1.0), ratio=(3 / 4, 4 / 3)):
    area = w * h
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target = area * rng.uniform(*scale)
        aspect = math.exp(rng.uniform(*log_ratio))
        cw = int(round(math.sqrt(target * aspect)))
        ch = int(round(math.sqrt(target / aspect)))
        if 0 < cw <= w and 0 < ch <= h:
            x = rng.randint(0, w - cw)
            y = rng.randint(0, h - ch)
            return x, y, cw, ch
    in_ratio = w / h
    if in_ratio < ratio[0]:
        cw, ch = w, int(round(w / ratio[0]))
    elif in_ratio > ratio[1]:
        cw, ch = int(round(h * ratio[1])), h
    else:
        cw, ch = w, h
    return (w - cw) // 2, (h - ch) // 2, cw, ch


class TrainTransform:
    """RandomResizedCrop(size) + random horizontal flip -> uint8 HWC array.

    The generator is private to the transform and must be given its own seed in every process
    that draws from it (:meth:`reseed`; the loader does that per (seed, rank, worker) --
    forked loader workers would otherwise all continue the parent's stream and cut identical
    crops)."""

    def __init__(self, size: int = 224, seed: int = 0) -> None:
        self.size = size
        self.rng = random.Random(seed)

    def reseed(self, seed: int) -> None:
        self.rng = random.Random(seed)

    def __call__(self, img):
        from PIL import Image

        w, h = img.size
        x, y, cw, ch = _random_resized_crop_box(w, h, self.rng)
        img = img.resize((self.size, self.size), Image.BILINEAR, box=(x, y, x + cw, y + ch))
        arr = np.array(img, dtype=np.uint8)
        if self.rng.random() < 0.5:
            arr = np.ascontiguousarray(arr[:, ::-1, :])
        return arr


class ValTransform:
    """Resize(shorter side = size * 256/224) + CenterCrop(size) -> uint8 HWC array."""

    def __init__(self, size: int = 224) -> None:
        self.size = size
        self.resize = int(round(size * 256 / 224))

    def __call__(self, img):
        from PIL import Image

        w, h = img.size
        if w <= h:
            nw, nh = self.resize, max(int(round(h * self.resize / w)), self.size)
        else:
            nw, nh = max(int(round(w * self.resize / h)), self.size), self.resize
        img = img.resize((nw, nh), Image.BILINEAR)
        left, top = (nw - self.size) // 2, (nh - self.size) // 2
        img = img.crop((left, top, left + self.size, top + self.size))
        return np.array(img, dtype=np.uint8)


def to_output_format(arr_hwc_u8: np.ndarray, fmt: str) -> torch.Tensor:
    t = torch.from_numpy(arr_hwc_u8)
    if fmt == "uint8_nhwc":
        return t
    x = t.permute(2, 0, 1).to(torch.float32).div_(255.0)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (x - mean) / std


# ----------------------------------------------------------------------------- datasets
class ImageFolderDataset(Dataset):
    """``root/<split>/<class_dir>/<image>``; classes are the sorted directory names."""

    def __init__(self, root: str, split: str, image_size: int = 224, fmt: str = "float_nchw") -> None:
        self.dir = os.path.join(os.path.expanduser(root), split)
        if not os.path.isdir(self.dir):
            raise FileNotFoundError(f"ImageNet split directory not found: {self.dir}")
        self.fmt = fmt
        classes = sorted(d.name for d in os.scandir(self.dir) if d.is_dir())
        if not classes:
            raise RuntimeError(f"no class directories under {self.dir}")
        self.class_to_idx = {c: i for i, c in enumerate(classes)}
        self.samples: List[Tuple[str, int]] = []
        for c in classes:
            cdir = os.path.join(self.dir, c)
            for fn in sorted(os.listdir(cdir)):
                if fn.lower().endswith(_IMG_EXT):
                    self.samples.append((os.path.join(cdir, fn), self.class_to_idx[c]))
        self.transform: Callable = TrainTransform(image_size) if split == "train" else ValTransform(image_size)

    def __len__(self) -> int:
        return len(self.samples)

    def __getitem__(self, i: int):
        from PIL import Image

        path, label = self.samples[i]
        with Image.open(path) as im:
            arr = self.transform(im.convert("RGB"))
        return to_output_format(arr, self.fmt), label


class SyntheticImageNet(Dataset):
    """Deterministic pseudo-random images and labels with ImageNet's shapes.

    Sample ``i`` is a pure function of ``(seed, split, i)``.  A small pool of pre-generated
    images is indexed by ``i`` so that item access is a cheap copy (the loader is then never
    the bottleneck of a benchmark); labels are ``hash(i) mod n_classes``.  With
    ``learnable=True`` the image mean encodes the label so a short run shows the loss
    falling (used by the integration test).
    """

    def __init__(self, split: str, length: int, n_classes: int = 1000, image_size: int = 224,
                 fmt: str = "float_nchw", seed: int = 0, pool: int = 64, learnable: bool = False) -> None:
        self.length = int(length)
        self.n_classes = int(n_classes)
        self.size = int(image_size)
        self.fmt = fmt
        self.learnable = learnable
        self.offset = 0 if split == "train" else 1_000_003
        g = torch.Generator().manual_seed(seed + (17 if split == "train" else 29))
        self.pool = torch.randint(0, 256, (pool, self.size, self.size, 3), generator=g, dtype=torch.uint8)

    def __len__(self) -> int:
        return self.length

    def label_of(self, i: int) -> int:
        return ((i + self.offset) * 2654435761 % (2 ** 32)) % self.n_classes

    def __getitem__(self, i: int):
        if not 0 <= i < self.length:
            raise IndexError(i)
        label = self.label_of(i)
        arr = self.pool[(i + self.offset) % self.pool.shape[0]]
        if self.learnable:
            # shift brightness by class so that the task is trivially learnable
            shift = int(255 * label / max(self.n_classes - 1, 1))
            arr = (arr.to(torch.int16) // 4 + int(0.75 * shift)).clamp_(0, 255).to(torch.uint8)
        if self.fmt == "uint8_nhwc":
            return arr.clone(), label
        return to_output_format(arr.numpy(), self.fmt), label


def get_dataset(name: str, root: str, split: str, n_classes: int = 1000, image_size: int = 224,
                fmt: str = "float_nchw", length: Optional[int] = None, seed: int = 0, **kw) -> Dataset:
    """``name in {"imagenet", "packed", "synthetic", "synthetic_learnable"}``, ``split in {"train", "val"}``.

    ``packed``: ``root`` holds the output of ``python -m pytorch_distributed_training_amd.data.packed``
    (fixed-size uint8 records, augmentation on the GPU unless ``cpu_augment=True``)."""
    if split not in ("train", "val"):
        raise ValueError(f"split must be 'train' or 'val', got '{split}'")
    lname = name.lower()
    if lname == "imagenet":
        return ImageFolderDataset(root, split, image_size, fmt)
    if lname == "packed":
        from .packed import PackedImageDataset
        return PackedImageDataset(root, split, image_size, fmt, cpu_augment=bool(kw.get("cpu_augment", False)), seed=seed)
    if lname in ("synthetic", "synthetic_learnable"):
        if length is None:
            length = 1281167 if split == "train" else 50000
        if lname == "synthetic_learnable":
            kw["learnable"] = True       # brightness encodes the label: a convergence check without data files
        return SyntheticImageNet(split, length, n_classes, image_size, fmt, seed, **kw)
    raise ValueError(f"unknown dataset '{name}' (imagenet | packed | synthetic | synthetic_learnable)")
