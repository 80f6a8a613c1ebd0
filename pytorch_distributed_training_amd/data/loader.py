"""Loaders: infinite iteration, per-rank sharding and device prefetch.

Upstream wraps its training ``DataLoader`` in ``make_iter_dataloader`` -- an endless
generator that restarts the loader at the end of each epoch -- and moves every batch to
the GPU with ``non_blocking=True`` from pinned memory at the top of the step
(reference ``train_distributed.py:227-249,252,272-273``).

:class:`DevicePrefetcher` additionally issues the host-to-device copy of batch *k+1* on a
dedicated HIP stream while step *k* computes, so the copy never sits on the critical
path; the consumer stream waits on an event, not on the host.
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import torch
from torch.utils.data import DataLoader, Dataset

from ..parallel import streams
from ..parallel.sampler import DistributedShardSampler


def make_iter_dataloader(loader: DataLoader, start_epoch: int = 0, skip_batches: int = 0) -> Iterator:
    """Endless ``(img, label)`` generator; advances the sampler's epoch at every restart.

    ``start_epoch`` / ``skip_batches`` position the stream when training resumes from a
    checkpoint taken in the middle of an epoch (the skipped samples are never loaded).
    """
    epoch = start_epoch
    skip = skip_batches * (loader.batch_size or 1)
    while True:
        sampler = getattr(loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            try:
                sampler.set_epoch(epoch, skip)
            except TypeError:          # torch's DistributedSampler: no mid-epoch positioning
                sampler.set_epoch(epoch)
        skip = 0
        for batch in loader:
            yield batch
        epoch += 1


def augmentation_seed(seed: int, rank: int, worker_id: int) -> int:
    """Distinct, reproducible stream per (run seed, rank, loader worker); ``worker_id`` -1 is the
    in-process loader (``num_workers == 0``)."""
    return (int(seed) * 1000003 + int(rank) * 7919 + int(worker_id) + 1) & 0x7FFFFFFFFFFFFFFF


class _SeedWorker:
    """``worker_init_fn`` (a picklable object: loader workers may be spawned): gives the dataset's
    random transform of this worker its own stream."""

    def __init__(self, seed: int, rank: int) -> None:
        self.seed, self.rank = seed, rank

    def __call__(self, worker_id: int) -> None:
        info = torch.utils.data.get_worker_info()
        transform = getattr(info.dataset, "transform", None) if info is not None else None
        if hasattr(transform, "reseed"):
            transform.reseed(augmentation_seed(self.seed, self.rank, worker_id))


def build_loader(dataset: Dataset, batch_size: int, num_workers: int, train: bool, world_size: int, rank: int,
                 seed: int = 0, pin_memory: bool = True, pad_val: bool = False, native: bool = True):
    """Per-rank loader: shuffled + ``drop_last`` shards for training, ordered shards for validation.  Packed training
    splits whose crop runs on the GPU get the native batch gatherer (``num_workers`` = its thread count, at least one)
    unless ``native=False``; everything else a ``torch.utils.data.DataLoader``."""
    sampler = DistributedShardSampler(
        len(dataset), num_replicas=world_size, rank=rank, shuffle=train, seed=seed,
        drop_last=train, pad=pad_val if not train else False,
    )
    transform = getattr(dataset, "transform", None)
    if hasattr(transform, "reseed"):
        transform.reseed(augmentation_seed(seed, rank, -1))
    if train and native and getattr(dataset, "gpu_augment", False) and hasattr(dataset, "native_records"):
        # whole stored records, cropped on the GPU: batches are gathered by native threads (see .native)
        from . import native as native_mod
        if native_mod.available():
            return native_mod.NativeRecordLoader(dataset, batch_size, sampler, threads=max(1, num_workers), drop_last=True,
                                                 pin=pin_memory)
    return DataLoader(
        dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
        pin_memory=pin_memory and torch.cuda.is_available(), drop_last=train,
        persistent_workers=num_workers > 0, worker_init_fn=_SeedWorker(seed, rank),
    )


class DevicePrefetcher:
    """Wraps a batch iterator; keeps one batch in flight to ``device`` on a copy stream.

    ``transform(img, label) -> (img, label)`` (optional) runs on that stream right behind the copy, i.e. under
    the previous step's compute: the GPU-side augmentation of packed datasets (``data.packed.GpuAugment``)."""

    def __init__(self, it: Iterator, device: torch.device, transform=None) -> None:
        self.it = iter(it)
        self.transform = transform
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        # one of the rank's four streams (parallel/streams.py): an upload queued behind kernels is measurably worse
        self.stream: Optional[torch.cuda.Stream] = streams.get("copy", self.device) if self.cuda else None
        self._next: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._event = None
        self._preload()

    def _preload(self) -> None:
        try:
            img, label = next(self.it)
        except StopIteration:
            self._next = None
            return
        if not self.cuda:
            self._next = (img, label) if self.transform is None else self.transform(img, label)
            return
        with torch.cuda.stream(self.stream):
            release = getattr(img, "pdt_release", None)
            img = img.to(self.device, non_blocking=True)
            label = label.to(self.device, non_blocking=True)
            if release is not None:                  # a loader that reuses its pinned buffers: tell it when the copy is done
                copied = torch.cuda.Event()
                copied.record(self.stream)
                release(copied)
            if self.transform is not None:
                img, label = self.transform(img, label)
            self._event = torch.cuda.Event()
            self._event.record(self.stream)
        self._next = (img, label)

    def __iter__(self) -> "DevicePrefetcher":
        return self

    def __next__(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._next is None:
            raise StopIteration
        img, label = self._next
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self._event)
            img.record_stream(cur)
            label.record_stream(cur)
        self._preload()
        return img, label
