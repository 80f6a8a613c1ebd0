"""Classification metrics and running meters.

``accuracy`` and ``AverageMeter`` reproduce the contracts the upstream trainer relies on
(reference ``train_distributed.py:32,305-321``): ``accuracy(pred, label, topk)`` returns
device tensors (they are passed to ``dist.all_reduce``), and ``AverageMeter.update(val)``
takes a single value and ``.value()`` returns the running mean.

``TopKCounter`` is the exact-count replacement used by this trainer's validation loop:
it keeps ``[loss_sum, correct@1, correct@5, seen]`` in ONE device tensor, so a whole
validation costs one all-reduce and one host synchronisation instead of three
all-reduces per batch (reference ``train_distributed.py:315-318``).
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch


def topk_correct(pred: torch.Tensor, label: torch.Tensor, topk: Sequence[int] = (1, 5)) -> Tuple[torch.Tensor, ...]:
    """Number of rows whose label is among the k largest logits, per k (0-dim tensors).

    Uses the rank formulation ``rank = #{j : logit_j > logit_label}`` so no sort / top-k
    selection is needed; ties are resolved in favour of the label, like a stable sort.
    """
    pred = pred.float()
    label = label.long()
    target_logit = pred.gather(1, label.view(-1, 1))
    rank = (pred > target_logit).sum(dim=1)
    return tuple((rank < k).sum().to(torch.float32) for k in topk)


def accuracy(pred: torch.Tensor, label: torch.Tensor, topk: Sequence[int] = (1, 5)) -> Tuple[torch.Tensor, ...]:
    """Top-k accuracy in percent, one 0-dim device tensor per k."""
    n = max(int(label.numel()), 1)
    return tuple(c * (100.0 / n) for c in topk_correct(pred, label, topk))


class AverageMeter:
    """Running mean of the values passed to :meth:`update` (tensors stay on device)."""

    def __init__(self) -> None:
        self.reset()

    def reset(self) -> None:
        self._sum = None
        self._count = 0

    def update(self, val, n: int = 1) -> None:
        if isinstance(val, torch.Tensor):
            val = val.detach()
        contrib = val * n
        self._sum = contrib if self._sum is None else self._sum + contrib
        self._count += n

    def value(self):
        if self._count == 0:
            return 0.0
        avg = self._sum / self._count
        return avg.item() if isinstance(avg, torch.Tensor) else float(avg)

    @property
    def count(self) -> int:
        return self._count


class TopKCounter:
    """On-device accumulator of ``[loss_sum, correct@1, correct@5, n_seen]``."""

    def __init__(self, device: torch.device) -> None:
        self.state = torch.zeros(4, dtype=torch.float64 if device.type == "cpu" else torch.float32, device=device)

    def update(self, loss_sum: torch.Tensor, correct1: torch.Tensor, correct5: torch.Tensor, n: int) -> None:
        self.state[0] += loss_sum.to(self.state.dtype)
        self.state[1] += correct1.to(self.state.dtype)
        self.state[2] += correct5.to(self.state.dtype)
        self.state[3] += n

    def add_tensor(self, t: torch.Tensor) -> None:
        """Add a ``[loss_sum, c1, c5, n]`` tensor produced by a fused metrics kernel."""
        self.state += t.to(self.state.dtype)

    def all_reduce(self, group=None) -> None:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            buf = self.state.to(torch.float32) if self.state.dtype == torch.float64 and self.state.is_cuda else self.state
            from ..parallel import collectives as coll
            coll.all_reduce(buf, group=group)
            if buf is not self.state:
                self.state.copy_(buf)

    def result(self) -> dict:
        loss_sum, c1, c5, n = [float(x) for x in self.state.tolist()]
        n = max(n, 1.0)
        return {"loss": loss_sum / n, "acc1": 100.0 * c1 / n, "acc5": 100.0 * c5 / n, "count": int(n)}
