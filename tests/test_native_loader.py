"""The native batch gatherer (data/csrc/batcher.cpp) and the loader built on it, against the torch DataLoader path."""
import os

import numpy as np
import pytest
import torch

from pytorch_distributed_training_amd.data import build_loader, make_iter_dataloader
from pytorch_distributed_training_amd.data import native
from pytorch_distributed_training_amd.data.packed import IMAGES, LABELS, META, PackedImageDataset

pytestmark = pytest.mark.skipif(not native.available(), reason="no C++ compiler for the host library")


def _packed_split(root, n=53, side=16, classes=5):
    import json
    d = os.path.join(root, "train")
    os.makedirs(d)
    rs = np.random.RandomState(0)
    imgs = rs.randint(0, 256, (n, side, side, 3), dtype=np.uint8)
    imgs.tofile(os.path.join(d, IMAGES))
    labels = rs.randint(0, classes, n).astype(np.int64)
    np.save(os.path.join(d, LABELS), labels)
    with open(os.path.join(d, META), "w") as f:
        json.dump({"n": n, "side": side, "classes": [f"c{i}" for i in range(classes)]}, f)
    return imgs, labels


def test_gatherer_copies_the_requested_records(tmp_path):
    imgs, _ = _packed_split(str(tmp_path))
    g = native.RecordGatherer(os.path.join(tmp_path, "train", IMAGES), len(imgs), imgs[0].size, threads=3)
    try:
        rs = np.random.RandomState(1)
        tickets = []
        for _ in range(6):                               # several jobs in flight, repeated and unordered indices
            idx = rs.randint(0, len(imgs), 37)
            dst = torch.zeros(37, *imgs.shape[1:], dtype=torch.uint8)
            tickets.append((g.submit(idx, dst), idx, dst))
        for t, idx, dst in reversed(tickets):            # waited for out of order
            g.wait(t)
            assert np.array_equal(dst.numpy(), imgs[idx])
        bad = g.submit(np.array([0, len(imgs)]), torch.zeros(2, *imgs.shape[1:], dtype=torch.uint8))
        with pytest.raises(IndexError):
            g.wait(bad)
        with pytest.raises(ValueError):
            g.submit(np.arange(4), torch.zeros(3, *imgs.shape[1:], dtype=torch.uint8))
    finally:
        g.close()


@pytest.mark.parametrize("world,rank", [(1, 0), (2, 1)])
def test_native_loader_yields_what_the_dataloader_yields(tmp_path, world, rank):
    imgs, labels = _packed_split(str(tmp_path))
    ds = PackedImageDataset(str(tmp_path), "train", image_size=8)
    fast = build_loader(ds, 6, 3, True, world, rank, seed=4, pin_memory=False)
    slow = build_loader(ds, 6, 0, True, world, rank, seed=4, pin_memory=False, native=False)
    assert isinstance(fast, native.NativeRecordLoader) and isinstance(slow, torch.utils.data.DataLoader)
    assert len(fast) == len(slow)
    a, b = make_iter_dataloader(fast, 1, 2), make_iter_dataloader(slow, 1, 2)       # resumed mid-epoch, across epochs
    for _ in range(3 * len(fast)):
        (xa, ya), (xb, yb) = next(a), next(b)
        assert torch.equal(xa.clone(), xb) and torch.equal(ya.clone(), yb)
    a.close()
    fast.close()


def test_buffers_are_not_reused_while_the_consumer_holds_them(tmp_path):
    imgs, labels = _packed_split(str(tmp_path), n=64)
    ds = PackedImageDataset(str(tmp_path), "train", image_size=8)
    loader = build_loader(ds, 4, 2, True, 1, 0, seed=0, pin_memory=False)
    it = iter(loader)
    first, second = next(it), next(it)
    keep = (first[0].clone(), second[0].clone())
    next(it)                                             # the consumer holds two batches while a third is produced
    assert torch.equal(first[0], keep[0]) and torch.equal(second[0], keep[1])
    it.close()                                           # abandoned mid-epoch: pending gathers are drained
    loader.close()
