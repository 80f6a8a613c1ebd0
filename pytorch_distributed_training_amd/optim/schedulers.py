"""Iteration-based learning-rate schedules with optional warm-up.

Upstream obtains its scheduler from ``get_scheduler(optimizer, cfg)``, steps it after
EVERY iteration and reads ``get_last_lr()`` for the log line; milestones are therefore
counted in iterations (reference ``train_distributed.py:211,285,299`` and
``config/ResNet50.yml:12-18``, where ``warmup_iters / warmup_mode / warmup_factor`` appear
as commented keys).

Schedules are closed-form functions of the iteration index, so resuming from a
checkpoint only needs the iteration counter.  They work with any optimizer object that
exposes ``param_groups`` (``torch.optim`` optimizers and this package's ``FlatSGD``).
"""
from __future__ import annotations

import bisect
import math
from typing import Any, Dict, List, Sequence


class IterLRScheduler:
    """Base: ``lr_i(t) = base_lr_i * warmup(t) * factor(t)``; call :meth:`step` once per iteration."""

    def __init__(self, optimizer, warmup_iters: int = 0, warmup_mode: str = "linear",
                 warmup_factor: float = 1.0 / 3.0, last_iter: int = 0) -> None:
        if warmup_mode not in ("linear", "constant"):
            raise ValueError(f"unknown warmup_mode '{warmup_mode}' (linear | constant)")
        self.optimizer = optimizer
        self.warmup_iters = int(warmup_iters or 0)
        self.warmup_mode = warmup_mode
        self.warmup_factor = float(warmup_factor)
        self.base_lrs: List[float] = []
        for group in optimizer.param_groups:
            group.setdefault("initial_lr", group["lr"])
            self.base_lrs.append(float(group["initial_lr"]))
        self.last_iter = int(last_iter)
        self._apply()

    # -- schedule -----------------------------------------------------------------
    def factor(self, it: int) -> float:  # pragma: no cover - abstract
        raise NotImplementedError

    def warmup(self, it: int) -> float:
        if self.warmup_iters <= 0 or it >= self.warmup_iters:
            return 1.0
        if self.warmup_mode == "constant":
            return self.warmup_factor
        alpha = it / float(self.warmup_iters)
        return self.warmup_factor * (1.0 - alpha) + alpha

    def lr_at(self, it: int) -> List[float]:
        scale = self.warmup(it) * self.factor(it)
        return [base * scale for base in self.base_lrs]

    # -- torch-like API -----------------------------------------------------------
    def _apply(self) -> None:
        self._last_lr = self.lr_at(self.last_iter)
        for group, lr in zip(self.optimizer.param_groups, self._last_lr):
            group["lr"] = lr

    def step(self) -> None:
        self.last_iter += 1
        self._apply()

    def get_last_lr(self) -> List[float]:
        return list(self._last_lr)

    def state_dict(self) -> Dict[str, Any]:
        return {"last_iter": self.last_iter, "base_lrs": list(self.base_lrs)}

    def load_state_dict(self, state: Dict[str, Any]) -> None:
        self.last_iter = int(state["last_iter"])
        if "base_lrs" in state and len(state["base_lrs"]) == len(self.base_lrs):
            self.base_lrs = [float(x) for x in state["base_lrs"]]
        self._apply()


class MultiStepLR(IterLRScheduler):
    """Multiply by ``gamma`` at each milestone iteration (upstream's ``multi_step``)."""

    def __init__(self, optimizer, milestones: Sequence[int], gamma: float = 0.1, **kw) -> None:
        self.milestones = sorted(int(m) for m in milestones)
        self.gamma = float(gamma)
        super().__init__(optimizer, **kw)

    def factor(self, it: int) -> float:
        return self.gamma ** bisect.bisect_right(self.milestones, it)


class CosineLR(IterLRScheduler):
    """Half-cosine from the base LR to ``min_lr`` over ``max_iters`` iterations."""

    def __init__(self, optimizer, max_iters: int, min_lr: float = 0.0, **kw) -> None:
        self.max_iters = max(int(max_iters), 1)
        self.min_lr = float(min_lr)
        super().__init__(optimizer, **kw)

    def lr_at(self, it: int) -> List[float]:
        t = min(it, self.max_iters) / self.max_iters
        cos = 0.5 * (1.0 + math.cos(math.pi * t))
        w = self.warmup(it)
        return [w * (self.min_lr + (base - self.min_lr) * cos) for base in self.base_lrs]

    def factor(self, it: int) -> float:  # unused (lr_at overridden)
        return 1.0


class PolyLR(IterLRScheduler):
    """``(1 - t/max_iters) ** power`` decay."""

    def __init__(self, optimizer, max_iters: int, power: float = 0.9, min_lr: float = 0.0, **kw) -> None:
        self.max_iters = max(int(max_iters), 1)
        self.power = float(power)
        self.min_lr = float(min_lr)
        super().__init__(optimizer, **kw)

    def lr_at(self, it: int) -> List[float]:
        t = min(it, self.max_iters) / self.max_iters
        f = (1.0 - t) ** self.power
        w = self.warmup(it)
        return [w * (self.min_lr + (base - self.min_lr) * f) for base in self.base_lrs]

    def factor(self, it: int) -> float:
        return 1.0


class ConstantLR(IterLRScheduler):
    def factor(self, it: int) -> float:
        return 1.0


def get_scheduler(optimizer, cfg: Dict[str, Any], train_iters: int = None) -> IterLRScheduler:
    """Build a scheduler from the ``training.lr_schedule`` config section."""
    cfg = dict(cfg)
    name = str(cfg.get("name", "multi_step")).lower()
    warm = {
        "warmup_iters": cfg.get("warmup_iters", 0) or 0,
        "warmup_mode": cfg.get("warmup_mode", "linear") or "linear",
        "warmup_factor": cfg.get("warmup_factor", 1.0 / 3.0),
    }
    max_iters = cfg.get("max_iters") or train_iters
    if name in ("multi_step", "multistep", "step"):
        return MultiStepLR(optimizer, milestones=cfg["milestones"], gamma=cfg.get("gamma", 0.1), **warm)
    if name == "cosine":
        if not max_iters:
            raise ValueError("cosine schedule needs max_iters (or train_iters)")
        return CosineLR(optimizer, max_iters=max_iters, min_lr=cfg.get("min_lr", 0.0) or 0.0, **warm)
    if name == "poly":
        if not max_iters:
            raise ValueError("poly schedule needs max_iters (or train_iters)")
        return PolyLR(optimizer, max_iters=max_iters, power=cfg.get("power", 0.9),
                      min_lr=cfg.get("min_lr", 0.0) or 0.0, **warm)
    if name == "constant":
        return ConstantLR(optimizer, **warm)
    raise ValueError(f"unknown lr schedule '{name}' (multi_step | cosine | poly | constant)")
