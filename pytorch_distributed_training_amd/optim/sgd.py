"""SGD with momentum and weight decay over flat buffers (one fused kernel per step).

Upstream instantiates ``torch.optim.SGD(model.parameters(), lr, weight_decay, momentum)``
from the ``training.optimizer`` config section through ``get_optimizer`` -- a single
parameter group, so weight decay also applies to BatchNorm weights/biases and the
classifier bias (reference ``train_distributed.py:204-207``, ``config/ResNet50.yml:7-11``).
Update rule (``torch/optim/sgd.py:322-367``): ``g += wd*p; buf = mu*buf + (1-damp)*g``
(first step: ``buf = g``); ``p -= lr*buf``.

:class:`FlatSGD` applies that rule to a :class:`~..engine.params.FlatParams` store with
one launch; the data-parallel ``1/world`` gradient average is folded into the same pass
(``grad_scale``), the gradients are zeroed for the next accumulation and the bf16 compute
mirror is rewritten.  It exposes ``param_groups`` so the LR schedulers work unchanged.
"""
from __future__ import annotations

from typing import Any, Dict, List, Tuple

import torch


class FlatSGD:
    def __init__(self, params, ops, lr: float, momentum: float = 0.0, weight_decay: float = 0.0,
                 nesterov: bool = False, dampening: float = 0.0, no_wd_bn: bool = False) -> None:
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        self.params = params
        self.ops = ops
        self.param_groups: List[Dict[str, Any]] = [{
            "lr": float(lr), "momentum": float(momentum), "weight_decay": float(weight_decay),
            "nesterov": bool(nesterov), "dampening": float(dampening),
        }]
        self.no_wd_bn = bool(no_wd_bn)
        self.steps = 0
        self._segments = self._build_segments()

    def _build_segments(self) -> List[Tuple[int, int, bool]]:
        """Contiguous ``(start, end, decay)`` runs; one run when decay applies everywhere."""
        if not self.no_wd_bn:
            return [(0, self.params.total, True)]
        segs: List[Tuple[int, int, bool]] = []
        for key in self.params.order:
            e = self.params.entries[key]
            end = self.params.end_offset(key)
            decay = not e.no_decay
            if segs and segs[-1][2] == decay:
                segs[-1] = (segs[-1][0], end, decay)
            else:
                segs.append((e.offset, end, decay))
        return segs

    def step(self, grad_scale: float = 1.0) -> None:
        self.step_range(0, self.params.total, grad_scale)
        self.finish_step()

    def step_range(self, lo: int, hi: int, grad_scale: float = 1.0) -> None:
        """The update of the flat elements ``[lo, hi)`` only (one gradient bucket): the engine
        enqueues it behind that bucket's all-reduce while backward is still running.  A step is
        complete when every range has been updated and :meth:`finish_step` was called."""
        g = self.param_groups[0]
        P = self.params
        for seg_lo, seg_hi, decay in self._segments:
            start, end = max(lo, seg_lo), min(hi, seg_hi)
            if start >= end:
                continue
            pc = None if P.pc is P.p else P.pc[start:end]
            self.ops.sgd_step(
                P.p[start:end], P.g[start:end], P.m[start:end], pc,
                g["lr"], g["momentum"], g["weight_decay"] if decay else 0.0, grad_scale,
                g["nesterov"], g["dampening"], self.steps == 0,
            )

    def finish_step(self) -> None:
        self.steps += 1

    def zero_grad(self, set_to_none: bool = True) -> None:  # gradients are zeroed inside step()
        pass

    def state_dict(self) -> Dict[str, Any]:
        return {
            "param_groups": [dict(g) for g in self.param_groups],
            "steps": self.steps,
            "momentum_buffers": self.params.export("m"),
        }

    def load_state_dict(self, state: Dict[str, Any]) -> None:
        for g, s in zip(self.param_groups, state["param_groups"]):
            g.update(s)
        self.steps = int(state.get("steps", 0))
        self.params.load_from(state["momentum_buffers"], which="m")

    def __repr__(self) -> str:
        g = self.param_groups[0]
        return (f"FlatSGD(lr={g['lr']}, momentum={g['momentum']}, weight_decay={g['weight_decay']}, "
                f"nesterov={g['nesterov']}, dampening={g['dampening']}, no_wd_bn={self.no_wd_bn}, "
                f"elements={self.params.total})")


_OPTIMIZERS = {"sgd": FlatSGD}


def get_optimizer(cfg: Dict[str, Any]):
    """Optimizer CLASS by ``cfg['name']`` (same contract as upstream's ``get_optimizer``)."""
    name = str(cfg["name"]).lower()
    if name not in _OPTIMIZERS:
        raise ValueError(f"unknown optimizer '{cfg['name']}'; available: {sorted(_OPTIMIZERS)}")
    return _OPTIMIZERS[name]
