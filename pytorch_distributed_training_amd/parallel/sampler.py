"""Distributed index sampler.

Upstream shards data with ``torch.utils.data.DistributedSampler`` -- training
``shuffle=True, drop_last=True``, validation ``shuffle=False`` (padded to a multiple of the
world size with repeated samples) -- and never calls ``set_epoch``, so every epoch replays
the same permutation (reference ``train_distributed.py:213-225``;
``torch/utils/data/distributed.py:107-124``).

This sampler keeps the partition rule (rank ``r`` takes indices ``r, r+W, r+2W, ...`` of
the epoch's permutation) and fixes the two quirks: the permutation depends on the epoch
(``set_epoch``, driven by the infinite iterator), and validation can run un-padded
(``pad=False``): ranks then see slightly different counts and the trainer sums exact
correct/seen counters instead of averaging padded per-batch accuracies.
"""
from __future__ import annotations

import math
from typing import Iterator, Optional

import torch
from torch.utils.data import Sampler


class DistributedShardSampler(Sampler):
    def __init__(self, dataset_len: int, num_replicas: int = 1, rank: int = 0, shuffle: bool = True,
                 seed: int = 0, drop_last: bool = False, pad: bool = True) -> None:
        if not 0 <= rank < num_replicas:
            raise ValueError(f"rank {rank} outside [0, {num_replicas})")
        self.n = int(dataset_len)
        self.world = int(num_replicas)
        self.rank = int(rank)
        self.shuffle = shuffle
        self.seed = int(seed)
        self.drop_last = drop_last
        self.pad = pad
        self.epoch = 0
        self.skip = 0            # samples of the NEXT iteration already consumed (mid-epoch resume)
        if drop_last:
            self.num_samples = self.n // self.world
        elif pad:
            self.num_samples = math.ceil(self.n / self.world)
        else:
            self.num_samples = len(range(self.rank, self.n, self.world))

    def set_epoch(self, epoch: int, skip_samples: int = 0) -> None:
        """Select the epoch's permutation; ``skip_samples`` drops that many leading samples of
        this rank's shard once (resuming in the middle of an epoch)."""
        self.epoch = int(epoch)
        self.skip = max(int(skip_samples), 0)

    def _permutation(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            return torch.randperm(self.n, generator=g).tolist()
        return list(range(self.n))

    def __iter__(self) -> Iterator[int]:
        idx = self._permutation()
        if self.drop_last:
            idx = idx[: self.num_samples * self.world]
        elif self.pad:
            total = self.num_samples * self.world
            if total > len(idx) and len(idx) > 0:
                reps = math.ceil((total - len(idx)) / len(idx))
                idx = idx + (idx * reps)[: total - len(idx)]
        shard = idx[self.rank::self.world]
        if self.skip:
            shard = shard[self.skip:]
            self.skip = 0
        return iter(shard)

    def __len__(self) -> int:
        return self.num_samples
