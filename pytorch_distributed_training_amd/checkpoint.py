"""Checkpoint save / load / resume.

Upstream has no checkpointing at all -- no ``torch.save`` / ``torch.load`` anywhere in
``train_distributed.py`` -- so a crash loses the run; only its ``.gitignore`` (``*.pth``,
``save/``) hints that weights were saved by hand (reference ``.gitignore:2,115``).

Format (a ``dict`` saved with ``torch.save``):

``model``      state_dict in the torchvision ResNet schema: fp32, OIHW, keys without a
               ``module.`` prefix, BN buffers incl. ``num_batches_tracked``; loads into
               ``torchvision.models.resnet50()`` (and into :class:`..models.resnet.ResNet`).
``optimizer``  hyper-parameters + momentum buffers keyed like ``model``.
``scheduler``  iteration counter / base LRs.
``iter``       next iteration to run.
``rng``        python / numpy / torch RNG states (tensors and plain containers only, so the
               file loads with ``torch.load(weights_only=True)``).
``config``     the validated configuration dict.

Files are written by rank 0 only, to a temporary name first and then renamed, so a reader
never sees a partial file.
"""
from __future__ import annotations

import logging
import os
import random
from typing import Any, Dict, List, Optional

import numpy as np
import torch


def _pack_rng() -> Dict[str, Any]:
    py = random.getstate()
    npst = np.random.get_state()
    out = {
        "python": [py[0], list(py[1]), py[2]],
        "numpy": {"kind": npst[0], "keys": torch.from_numpy(np.asarray(npst[1]).astype(np.int64)),
                  "pos": int(npst[2]), "has_gauss": int(npst[3]), "cached": float(npst[4])},
        "torch": torch.get_rng_state(),
    }
    if torch.cuda.device_count() > 0 and torch.cuda.is_available():
        out["cuda"] = [s for s in torch.cuda.get_rng_state_all()]
    return out


def _unpack_rng(state: Dict[str, Any], logger: Optional[logging.Logger] = None) -> bool:
    """Restore the saved generator states; ``False`` (and a logged warning naming the part that failed) when one of them
    could not be restored -- a resumed run then continues, but no longer reproduces the uninterrupted run's data order
    and augmentation stream from here on."""
    log = logger or logging.getLogger(__name__)
    ok = True

    def part(what: str, fn) -> None:
        nonlocal ok
        try:
            fn()
        except Exception as exc:                     # noqa: BLE001 -- one bad part must not stop the others
            ok = False
            log.warning("resume: the %s random-generator state of the checkpoint was NOT restored (%s: %s)", what,
                        type(exc).__name__, exc)

    def _py():
        py = state["python"]
        random.setstate((py[0], tuple(py[1]), py[2]))

    def _np():
        n = state["numpy"]
        np.random.set_state((n["kind"], n["keys"].numpy().astype(np.uint32), n["pos"], n["has_gauss"], n["cached"]))

    def _cuda():
        if not (torch.cuda.device_count() > 0 and torch.cuda.is_available()):
            return
        if torch.cuda.device_count() != len(state["cuda"]):
            raise RuntimeError(f"checkpoint holds {len(state['cuda'])} GPU generator states, this process sees "
                               f"{torch.cuda.device_count()} GPUs")
        torch.cuda.set_rng_state_all([s.cpu() for s in state["cuda"]])

    part("python", _py)
    part("numpy", _np)
    part("torch", lambda: torch.set_rng_state(state["torch"].cpu()))
    if "cuda" in state:
        part("GPU", _cuda)
    return ok


def _to_cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def atomic_save(obj: Any, path: str) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    tmp = f"{path}.tmp.{os.getpid()}"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def save_checkpoint(path: str, model_state: Dict[str, torch.Tensor], optimizer_state: Optional[Dict[str, Any]] = None,
                    scheduler_state: Optional[Dict[str, Any]] = None, iteration: int = 0,
                    config: Optional[Dict[str, Any]] = None, extra: Optional[Dict[str, Any]] = None) -> None:
    ckpt = {
        "model": _to_cpu(model_state),
        "optimizer": _to_cpu(optimizer_state) if optimizer_state is not None else None,
        "scheduler": scheduler_state,
        "iter": int(iteration),
        "rng": _pack_rng(),
        "config": config,
        "format": "torchvision-resnet-v1",
    }
    if extra:
        ckpt["extra"] = _to_cpu(extra)
    atomic_save(ckpt, path)


def save_model_only(path: str, model_state: Dict[str, torch.Tensor]) -> None:
    """Bare torchvision-style ``state_dict`` file (what ``model.load_state_dict`` expects)."""
    atomic_save(_to_cpu(model_state), path)


def load_checkpoint(path: str, map_location="cpu") -> Dict[str, Any]:
    """Load a training checkpoint or a bare state_dict (returned as ``{"model": ...}``)."""
    obj = torch.load(os.path.expanduser(path), map_location=map_location, weights_only=True)
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        return obj
    return {"model": obj, "optimizer": None, "scheduler": None, "iter": 0, "rng": None, "config": None}


def restore_rng(ckpt: Dict[str, Any], logger: Optional[logging.Logger] = None) -> bool:
    """``True`` when every generator state of the checkpoint was restored (or it holds none)."""
    if ckpt.get("rng"):
        return _unpack_rng(ckpt["rng"], logger)
    return True


def latest_checkpoint(directory: str) -> Optional[str]:
    """Newest ``iter_*.pth`` in ``directory`` (by iteration number), or ``None``."""
    if not directory or not os.path.isdir(directory):
        return None
    best, best_it = None, -1
    for fn in os.listdir(directory):
        if fn.startswith("iter_") and fn.endswith(".pth"):
            try:
                it = int(fn[5:-4])
            except ValueError:
                continue
            if it > best_it:
                best, best_it = os.path.join(directory, fn), it
    return best


def prune_checkpoints(directory: str, keep: int, upto: Optional[int] = None) -> List[str]:
    """Delete all but the ``keep`` newest periodic checkpoints (``iter_<n>.pth``) of ``directory``
    and return the removed paths.  Files with other names (explicitly tagged saves) stay.  A
    450 000-iteration run with a checkpoint every 2000 iterations would otherwise leave 225
    full training states (~300 MB each for ResNet-50) behind.

    ``upto``: the iteration just saved.  Only checkpoints up to it count: files with a HIGHER iteration number are
    left-overs of another run in a reused directory -- they are neither deleted nor allowed to push this run's own
    checkpoints out of the newest ``keep`` (see :func:`stale_checkpoints`)."""
    if keep <= 0 or not directory or not os.path.isdir(directory):
        return []
    found = []
    for fn in os.listdir(directory):
        if fn.startswith("iter_") and fn.endswith(".pth"):
            try:
                it = int(fn[5:-4])
            except ValueError:
                continue
            if upto is None or it <= upto:
                found.append((it, os.path.join(directory, fn)))
    found.sort()
    removed = []
    for _, path in found[:-keep]:
        try:
            os.remove(path)
            removed.append(path)
        except OSError:
            pass
    return removed


def stale_checkpoints(directory: str, current_iter: int) -> List[str]:
    """Periodic checkpoints of ``directory`` whose iteration lies beyond ``current_iter``: a fresh run pointed at a
    directory another run has used.  ``resume: auto`` would pick the newest of them."""
    out = []
    if directory and os.path.isdir(directory):
        for fn in os.listdir(directory):
            if fn.startswith("iter_") and fn.endswith(".pth"):
                try:
                    if int(fn[5:-4]) > current_iter:
                        out.append(os.path.join(directory, fn))
                except ValueError:
                    continue
    return sorted(out)
