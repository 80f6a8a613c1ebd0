"""Packed image store + GPU-side augmentation: an input path that keeps up with the GPU.

Upstream feeds its trainer from a JPEG ``ImageFolder`` through 16 ``DataLoader`` workers that decode,
crop and resize with PIL (reference ``train_distributed.py:227-249``, ``config/ResNet50.yml:23``).
One MI355X consumes >13 000 images/s on ResNet-50; a PIL worker delivers a few hundred, so that path
needs ~50-100 CPU cores per GPU.  This module moves the per-image work off the host:

* :func:`pack_image_folder` decodes the dataset ONCE into fixed-size records -- shorter side resized
  to ``side`` (256), centre square -- stored as one flat ``uint8 [N, side, side, 3]`` file plus labels
  (1.28 M images: 252 GB, which a training node holds in page cache);
* :class:`PackedImageDataset` memory-maps that file; a training item is a plain copy of one record
  (no decode, no resize), a validation item is the centre ``image_size`` crop of it (identical to
  Resize(256) + CenterCrop(224) when ``side = 256``);
* :class:`GpuAugment` draws the RandomResizedCrop box and the flip for every sample of a batch on the
  host (a seeded stream per run seed and rank) and the crop + bilinear resize + flip run in one HIP kernel
  on the copy stream, behind the batch's host-to-device copy (``ops.hip.crop_resize_flip``).

Difference to the PIL path, stated so that nobody has to find it: crops come from the stored centre
square (the strips a non-square image has beyond it are never seen), and the resize is plain bilinear
without PIL's antialiasing filter.
"""
from __future__ import annotations

import json
import math
import os
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

META = "meta.json"
IMAGES = "images.u8"
LABELS = "labels.npy"


def _load_square(path: str, side: int) -> np.ndarray:
    from PIL import Image

    with Image.open(path) as im:
        im.draft("RGB", (side, side))            # JPEG: decode at the smallest DCT scale that still covers `side`
        im = im.convert("RGB")
        w, h = im.size
        scale = side / min(w, h)
        nw, nh = max(side, int(round(w * scale))), max(side, int(round(h * scale)))
        im = im.resize((nw, nh), Image.BILINEAR)
        left, top = (nw - side) // 2, (nh - side) // 2
        return np.asarray(im.crop((left, top, left + side, top + side)), dtype=np.uint8)


def _pack_range(args) -> int:
    out_path, side, start, items = args
    mm = np.memmap(out_path, dtype=np.uint8, mode="r+")
    rec = side * side * 3
    for k, path in enumerate(items):
        mm[(start + k) * rec:(start + k + 1) * rec] = _load_square(path, side).reshape(-1)
    mm.flush()
    return len(items)


def pack_image_folder(root: str, split: str, out_dir: str, side: int = 256, workers: int = 8,
                      chunk: int = 256) -> str:
    """``root/<split>/<class>/<image>`` -> ``out_dir/<split>/{images.u8, labels.npy, meta.json}``."""
    from .datasets import ImageFolderDataset

    src = ImageFolderDataset(root, split, side, "uint8_nhwc")
    dst = os.path.join(out_dir, split)
    os.makedirs(dst, exist_ok=True)
    n = len(src.samples)
    path = os.path.join(dst, IMAGES)
    np.memmap(path, dtype=np.uint8, mode="w+", shape=(n * side * side * 3,)).flush()
    jobs = [(path, side, s, [p for p, _ in src.samples[s:s + chunk]]) for s in range(0, n, chunk)]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp

        with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:
            done = sum(pool.imap_unordered(_pack_range, jobs))
    else:
        done = sum(_pack_range(j) for j in jobs)
    assert done == n
    np.save(os.path.join(dst, LABELS), np.asarray([l for _, l in src.samples], dtype=np.int64))
    classes = sorted(src.class_to_idx, key=src.class_to_idx.get)
    with open(os.path.join(dst, META), "w") as f:
        json.dump({"n": n, "side": side, "classes": classes, "layout": "uint8 [n, side, side, 3]"}, f)
    return dst


class PackedImageDataset(Dataset):
    """Memory-mapped packed split.  ``split="train"`` items are whole stored records (``[side, side, 3]``
    uint8: the crop happens on the GPU, see :class:`GpuAugment`) unless ``cpu_augment`` asks for the crop on
    the host (a transform with ``reseed``, like the JPEG path); ``split="val"`` items are centre crops."""

    def __init__(self, root: str, split: str, image_size: int = 224, fmt: str = "uint8_nhwc",
                 cpu_augment: bool = False, seed: int = 0) -> None:
        self.dir = os.path.join(os.path.expanduser(root), split)
        meta_path = os.path.join(self.dir, META)
        if not os.path.exists(meta_path):
            raise FileNotFoundError(f"packed split not found: {meta_path} (build it with "
                                    "`python -m pytorch_distributed_training_amd.data.packed`)")
        with open(meta_path) as f:
            meta = json.load(f)
        self.n, self.side = int(meta["n"]), int(meta["side"])
        self.classes: List[str] = list(meta["classes"])
        self.size = int(image_size)
        if self.size > self.side:
            raise ValueError(f"image_size {self.size} exceeds the stored side {self.side}")
        self.fmt = fmt
        self.train = split == "train"
        self.labels = np.load(os.path.join(self.dir, LABELS))
        expect = self.n * self.side * self.side * 3
        if os.path.getsize(os.path.join(self.dir, IMAGES)) != expect or len(self.labels) != self.n:
            raise RuntimeError(f"{self.dir}: image file or labels do not match meta.json (n={self.n}, side={self.side})")
        self._mm: Optional[np.memmap] = None          # opened lazily: a memmap must not be shared through fork/pickle
        self.transform = _CpuCrop(self.size, seed) if (self.train and cpu_augment) else None
        self.gpu_augment = self.train and not cpu_augment
        if self.gpu_augment and fmt != "uint8_nhwc":
            raise ValueError("packed training records are raw uint8: use dataset.input_format = uint8_nhwc")

    def __len__(self) -> int:
        return self.n

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_mm"] = None
        return st

    def _images(self) -> np.memmap:
        if self._mm is None:
            self._mm = np.memmap(os.path.join(self.dir, IMAGES), dtype=np.uint8, mode="r",
                                 shape=(self.n, self.side, self.side, 3))
        return self._mm

    def native_records(self):
        """``(path, n, record shape, labels)`` of the training records as they lie in the file: what the native batch
        gatherer needs (:mod:`.native`).  Only whole-record items qualify (training split, crop on the GPU)."""
        if not self.gpu_augment:
            raise ValueError("only the GPU-augmented training split hands out whole records")
        return os.path.join(self.dir, IMAGES), self.n, (self.side, self.side, 3), self.labels

    def __getitem__(self, i: int):
        from .datasets import to_output_format

        rec = self._images()[i]
        label = int(self.labels[i])
        if self.train:
            if self.transform is None:
                return torch.from_numpy(np.array(rec)), label          # one copy out of the page cache
            return to_output_format(self.transform(rec), self.fmt), label
        o = (self.side - self.size) // 2
        return to_output_format(np.array(rec[o:o + self.size, o:o + self.size]), self.fmt), label


def random_resized_crop_boxes(n: int, side: int, rng: np.random.Generator, scale=(0.08, 1.0),
                              ratio=(3 / 4, 4 / 3)) -> np.ndarray:
    """``[n, 5]`` int32 ``(x, y, w, h, flip)``: torchvision's RandomResizedCrop sampling (ten tries of a random
    area fraction and log-uniform aspect ratio, then the centre fallback) on a ``side x side`` image, vectorised
    over the batch, plus a fair coin for the flip."""
    area = float(side * side)
    out = np.zeros((n, 5), dtype=np.int32)
    todo = np.ones(n, dtype=bool)
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1], n)
        aspect = np.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1]), n))
        cw = np.rint(np.sqrt(target * aspect)).astype(np.int64)
        ch = np.rint(np.sqrt(target / aspect)).astype(np.int64)
        ux, uy = rng.random(n), rng.random(n)
        ok = todo & (cw > 0) & (ch > 0) & (cw <= side) & (ch <= side)
        x = np.floor(ux * (side - cw + 1)).astype(np.int64)
        y = np.floor(uy * (side - ch + 1)).astype(np.int64)
        out[ok, 0], out[ok, 1], out[ok, 2], out[ok, 3] = x[ok], y[ok], cw[ok], ch[ok]
        todo &= ~ok
    out[todo, 0:4] = (0, 0, side, side)               # square image: the fallback is the whole image
    out[:, 4] = rng.random(n) < 0.5
    return out


class _CpuCrop:
    """Host-side twin of the GPU augmentation (same box sampling, PIL resize): A/B aid and CPU fallback."""

    def __init__(self, size: int, seed: int = 0) -> None:
        self.size = size
        self.rng = np.random.default_rng(seed)

    def reseed(self, seed: int) -> None:
        self.rng = np.random.default_rng(seed)

    def __call__(self, rec: np.ndarray) -> np.ndarray:
        from PIL import Image

        x, y, w, h, flip = random_resized_crop_boxes(1, rec.shape[0], self.rng)[0]
        im = Image.fromarray(np.asarray(rec)).resize((self.size, self.size), Image.BILINEAR, box=(x, y, x + w, y + h))
        arr = np.array(im, dtype=np.uint8)
        return np.ascontiguousarray(arr[:, ::-1]) if flip else arr


class GpuAugment:
    """Batch transform for :class:`..data.loader.DevicePrefetcher`: stored records ``[B, S, S, 3]`` (already on
    the device) -> training crops ``[B, size, size, 3]`` uint8.  Boxes come from a host stream seeded by
    (run seed, rank); the stream position is the number of batches drawn, so a resumed run continues it with
    :meth:`skip`."""

    def __init__(self, size: int, seed: int, rank: int, ops=None) -> None:
        from .loader import augmentation_seed

        self.size = int(size)
        self._seed = augmentation_seed(seed, rank, -2)
        self.rng = np.random.default_rng(self._seed)
        self.ops = ops
        self.batches = 0

    def skip(self, batches: int, batch_size: int, side: int) -> None:
        for _ in range(int(batches)):
            random_resized_crop_boxes(batch_size, side, self.rng)
        self.batches += int(batches)

    def __call__(self, img: torch.Tensor, label: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if img.dtype != torch.uint8 or img.dim() != 4 or img.shape[1] == self.size:
            return img, label                      # not a stored-record batch (validation, synthetic data)
        boxes = torch.from_numpy(random_resized_crop_boxes(img.shape[0], img.shape[1], self.rng))
        self.batches += 1
        ops = self.ops
        if ops is None:
            from ..ops import get_ops
            ops = self.ops = get_ops(img.device, "auto")
        return ops.crop_resize_flip(img, boxes, self.size), label


def main(argv=None) -> int:
    import argparse

    ap = argparse.ArgumentParser(description="Pack an ImageFolder split into fixed-size uint8 records")
    ap.add_argument("--root", required=True, help="dataset root holding <split>/<class>/<image>")
    ap.add_argument("--out", required=True)
    ap.add_argument("--split", action="append", default=None, help="train / val (repeatable; default both)")
    ap.add_argument("--side", type=int, default=256)
    ap.add_argument("--workers", type=int, default=8)
    args = ap.parse_args(argv)
    for split in args.split or ["train", "val"]:
        print(pack_image_folder(args.root, split, args.out, args.side, args.workers))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
