"""Multi-process text logging and rank-0 scalar logging.

Upstream funnels the log records of all worker processes through one ``mp.Queue`` into a
listener that owns the real logger (``MultiProcessLoggerListener(ctor, "spawn")`` with
``.queue``, ``.get_logger()``, ``.stop()``), builds that logger with
``get_train_logger(logdir, filename)``, gives every worker a ``worker_rank_{r}`` logger
with a ``QueueHandler`` and ``propagate=False``, and writes TensorBoard scalars on rank 0
only via ``get_tb_writer(log_dir, filename)``
(reference ``train_distributed.py:56-62,72-74,155-164,295-297,329-331``).

Implemented on the standard library: a spawn-context queue drained by a
``logging.handlers.QueueListener`` thread in the launcher process.  The scalar writer
always appends to ``<log_dir>/tf-board-logs/<name>/metrics.jsonl`` and mirrors into a
TensorBoard ``SummaryWriter`` when the ``tensorboard`` package is importable.
"""
from __future__ import annotations

import json
import logging
import logging.handlers
import multiprocessing as mp
import os
import sys
import time
from typing import Callable, Optional

_FMT = "%(asctime)s [%(name)s] %(levelname)s: %(message)s"


def get_train_logger(logdir: Optional[str], filename: Optional[str], name: str = "train") -> logging.Logger:
    """Logger writing to the console and to ``<logdir>/<filename>.log``."""
    logger = logging.getLogger(name)
    logger.setLevel(logging.INFO)
    logger.propagate = False
    for h in list(logger.handlers):
        logger.removeHandler(h)
        try:
            h.close()
        except Exception:
            pass
    fmt = logging.Formatter(_FMT)
    console = logging.StreamHandler(sys.stdout)
    console.setFormatter(fmt)
    logger.addHandler(console)
    if logdir:
        os.makedirs(logdir, exist_ok=True)
        path = os.path.join(logdir, f"{filename or 'train'}.log")
        fh = logging.FileHandler(path)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    return logger


class MultiProcessLoggerListener:
    """Owns the cross-process log queue and the thread that drains it."""

    def __init__(self, logger_constructor: Callable[[], logging.Logger], start_method: str = "spawn") -> None:
        self._ctx = mp.get_context(start_method)
        self.queue = self._ctx.Queue(-1)
        self._target = logger_constructor()
        self._listener = logging.handlers.QueueListener(self.queue, *self._target.handlers,
                                                        respect_handler_level=True)
        self._listener.start()
        self._stopped = False

    def get_logger(self, name: str = "main") -> logging.Logger:
        """Logger for the launcher process (records go through the same queue)."""
        return worker_logger(self.queue, name)

    def stop(self) -> None:
        if self._stopped:
            return
        self._stopped = True
        time.sleep(0.05)          # let in-flight records reach the queue
        try:
            self._listener.stop()
        finally:
            for h in self._target.handlers:
                try:
                    h.flush()
                except Exception:
                    pass
            try:
                self.queue.close()
                self.queue.join_thread()
            except Exception:
                pass


def worker_logger(queue, name: str, level: int = logging.INFO) -> logging.Logger:
    """Per-process logger that only forwards records into ``queue``."""
    logger = logging.getLogger(name)
    logger.propagate = False
    logger.setLevel(level)
    for h in list(logger.handlers):
        logger.removeHandler(h)
    if queue is not None:
        logger.addHandler(logging.handlers.QueueHandler(queue))
    else:
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter(_FMT))
        logger.addHandler(h)
    return logger


class ScalarWriter:
    """``add_scalar(tag, value, step)`` -> JSONL always, TensorBoard when available."""

    def __init__(self, log_dir: str) -> None:
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self._f = open(os.path.join(log_dir, "metrics.jsonl"), "a", buffering=1)
        self._tb = None
        try:  # optional dependency
            from torch.utils.tensorboard import SummaryWriter  # type: ignore

            self._tb = SummaryWriter(log_dir)
        except Exception:
            self._tb = None

    def add_scalar(self, tag: str, value, step: int) -> None:
        value = float(value)
        self._f.write(json.dumps({"tag": tag, "value": value, "step": int(step), "time": time.time()}) + "\n")
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)

    def add_scalars(self, values: dict, step: int) -> None:
        for tag, v in values.items():
            self.add_scalar(tag, v, step)

    def flush(self) -> None:
        self._f.flush()
        if self._tb is not None:
            self._tb.flush()

    def close(self) -> None:
        self.flush()
        self._f.close()
        if self._tb is not None:
            self._tb.close()


def get_tb_writer(log_dir: Optional[str], filename: Optional[str]) -> ScalarWriter:
    """Scalar writer under ``<log_dir>/tf-board-logs/<filename>``."""
    base = os.path.join(log_dir or ".", "tf-board-logs", filename or "run")
    return ScalarWriter(base)
