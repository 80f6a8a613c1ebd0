import numpy as np
import pytest
import torch

from pytorch_distributed_training_amd.data import DevicePrefetcher, build_loader, get_dataset, make_iter_dataloader
from pytorch_distributed_training_amd.parallel.sampler import DistributedShardSampler


def test_sampler_partition_properties():
    n, world = 103, 4
    shards = []
    for r in range(world):
        s = DistributedShardSampler(n, world, r, shuffle=True, seed=3, drop_last=True)
        shards.append(list(s))
        assert len(shards[-1]) == len(s) == n // world
    flat = sum(shards, [])
    assert len(set(flat)) == len(flat)                       # disjoint
    # padded validation shards cover everything, equal length
    shards = [list(DistributedShardSampler(n, world, r, shuffle=False, pad=True)) for r in range(world)]
    assert all(len(s) == 26 for s in shards) and set(sum(shards, [])) == set(range(n))
    # un-padded shards cover everything exactly once
    shards = [list(DistributedShardSampler(n, world, r, shuffle=False, pad=False)) for r in range(world)]
    assert sorted(sum(shards, [])) == list(range(n))


def test_sampler_epoch_changes_order():
    s = DistributedShardSampler(50, 2, 0, shuffle=True, seed=1, drop_last=True)
    s.set_epoch(0); a = list(s)
    s.set_epoch(1); b = list(s)
    s.set_epoch(0); c = list(s)
    assert a != b and a == c


@pytest.mark.parametrize("fmt", ["float_nchw", "uint8_nhwc"])
def test_synthetic_dataset(fmt):
    ds = get_dataset("synthetic", "", "train", n_classes=10, image_size=32, fmt=fmt, length=20, seed=1)
    img, label = ds[3]
    assert 0 <= label < 10 and len(ds) == 20
    if fmt == "float_nchw":
        assert img.shape == (3, 32, 32) and img.dtype == torch.float32
    else:
        assert img.shape == (32, 32, 3) and img.dtype == torch.uint8
    img2, label2 = ds[3]
    assert torch.equal(img, img2) and label == label2       # deterministic
    with pytest.raises(IndexError):
        ds[20]
    with pytest.raises(ValueError):
        get_dataset("synthetic", "", "test")


def test_infinite_iterator_and_prefetcher():
    ds = get_dataset("synthetic", "", "train", n_classes=10, image_size=16, length=10)
    loader = build_loader(ds, 4, 0, True, 1, 0, seed=0)
    assert len(loader) == 2                                   # drop_last
    it = DevicePrefetcher(make_iter_dataloader(loader), torch.device("cpu"))
    seen = [next(it)[1].tolist() for _ in range(5)]           # crosses two epoch boundaries
    assert all(len(b) == 4 for b in seen)
    assert seen[0] != seen[2]                                  # epoch-dependent shuffle


def test_imagefolder_dataset(tmp_path):
    from PIL import Image
    for split in ("train", "val"):
        for ci, cname in enumerate(("n01", "n02")):
            d = tmp_path / split / cname
            d.mkdir(parents=True)
            for j in range(3):
                arr = np.random.RandomState(ci * 10 + j).randint(0, 255, (40 + j, 50, 3), dtype=np.uint8)
                Image.fromarray(arr).save(d / f"img{j}.JPEG")
    tr = get_dataset("imagenet", str(tmp_path), "train", image_size=32)
    va = get_dataset("imagenet", str(tmp_path), "val", image_size=32, fmt="uint8_nhwc")
    assert len(tr) == 6 and len(va) == 6
    x, y = tr[0]
    assert x.shape == (3, 32, 32) and y == 0 and x.dtype == torch.float32
    xv, yv = va[5]
    assert xv.shape == (32, 32, 3) and xv.dtype == torch.uint8 and yv == 1
    with pytest.raises(FileNotFoundError):
        get_dataset("imagenet", str(tmp_path / "nope"), "train")
