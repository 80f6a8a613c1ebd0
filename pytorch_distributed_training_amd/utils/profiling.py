"""Step timers, throughput meter and optional roctx ranges.

Upstream has no profiling at all: its train log line carries only iteration, LR and loss
(reference ``train_distributed.py:286-294``).  This module adds (a) a throughput meter
used for the images/s, step-ms and data-wait-ms fields of the train log line and
``metrics.jsonl``, and (b) ``roctx`` ranges (``rocprofv3 --marker-trace`` shows them)
around the phases of a step, enabled by ``runtime.profile``.
"""
from __future__ import annotations

import contextlib
import ctypes
import time
from typing import Optional

import torch

_roctx = None
_roctx_tried = False


def _load_roctx():
    global _roctx, _roctx_tried
    if _roctx_tried:
        return _roctx
    _roctx_tried = True
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                 "/opt/rocm/lib/libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.restype = ctypes.c_int
            _roctx = lib
            break
        except (OSError, AttributeError):
            continue
    return _roctx


_enabled = False


def enable_ranges(flag: bool = True) -> None:
    global _enabled
    _enabled = bool(flag) and _load_roctx() is not None


@contextlib.contextmanager
def trace_range(name: str):
    """``with trace_range("backward"):`` -- a roctx range when profiling is enabled."""
    if _enabled:
        _roctx.roctxRangePushA(name.encode())
        try:
            yield
        finally:
            _roctx.roctxRangePop()
    else:
        yield


class ThroughputMeter:
    """Wall-clock step timing between log lines (host timers; one device sync per read)."""

    def __init__(self, device: Optional[torch.device] = None) -> None:
        self.device = device
        self.reset()

    def reset(self) -> None:
        self._t0 = None
        self._steps = 0
        self._samples = 0
        self._data_wait = 0.0

    def start(self) -> None:
        self._sync()
        self._t0 = time.perf_counter()
        self._steps = 0
        self._samples = 0
        self._data_wait = 0.0

    def add_step(self, n_samples: int, data_wait_s: float = 0.0) -> None:
        self._steps += 1
        self._samples += n_samples
        self._data_wait += data_wait_s

    def _sync(self) -> None:
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def read(self) -> dict:
        """Return averages since :meth:`start` and restart the window."""
        if self._t0 is None or self._steps == 0:
            self.start()
            return {"step_ms": 0.0, "images_per_s": 0.0, "data_wait_ms": 0.0}
        self._sync()
        dt = max(time.perf_counter() - self._t0, 1e-9)
        out = {
            "step_ms": 1e3 * dt / self._steps,
            "images_per_s": self._samples / dt,
            "data_wait_ms": 1e3 * self._data_wait / self._steps,
        }
        self._t0 = time.perf_counter()
        self._steps = 0
        self._samples = 0
        self._data_wait = 0.0
        return out


def peak_memory_gb(device: torch.device) -> float:
    if device.type == "cuda":
        return torch.cuda.max_memory_allocated(device) / 2 ** 30
    return 0.0
