"""Native batch assembly for packed datasets (``csrc/batcher.cpp`` -> ``libpdt_data.so``, plain ``g++``).

:class:`NativeRecordLoader` stands where a ``torch.utils.data.DataLoader`` stood for the packed training split: same
sampler, same batch order, same ``(img, label)`` items -- but a batch is gathered from the memory-mapped record file
straight into a pinned buffer by a pool of C++ threads (one host copy instead of four, no Python per record, no worker
processes).  Upstream's loader is the stock DataLoader with ``num_workers`` JPEG-decoding processes
(reference ``train_distributed.py:227-249``); the packed format exists because that path cannot feed an MI355X.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from collections import deque
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "batcher.cpp")
LIB = os.path.join(HERE, "libpdt_data.so")

_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile the host library in-tree (seconds); returns its path."""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cxx = os.environ.get("CXX", "g++")
    tmp = LIB + f".{os.getpid()}.tmp"
    subprocess.run([cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", SRC, "-o", tmp], check=True)
    os.replace(tmp, LIB)
    return LIB


def load():
    """The library (built on first use when the compiler is there); raises ``OSError`` / ``CalledProcessError`` if not."""
    global _lib
    with _lock:
        if _lib is None:
            lib = ctypes.CDLL(build())
            lib.pdt_batcher_open.restype = ctypes.c_void_p
            lib.pdt_batcher_open.argtypes = [ctypes.c_char_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int]
            lib.pdt_batcher_submit.restype = ctypes.c_longlong
            lib.pdt_batcher_submit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            lib.pdt_batcher_wait.restype = ctypes.c_int
            lib.pdt_batcher_wait.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
            lib.pdt_batcher_close.restype = None
            lib.pdt_batcher_close.argtypes = [ctypes.c_void_p]
            _lib = lib
    return _lib


def available() -> bool:
    try:
        load()
        return True
    except (OSError, subprocess.CalledProcessError, FileNotFoundError):
        return False


class RecordGatherer:
    """Thin handle on the C++ batcher: ``submit(indices, dst) -> ticket``, ``wait(ticket)``."""

    def __init__(self, path: str, n: int, rec_bytes: int, threads: int) -> None:
        self._lib = load()
        self._h = self._lib.pdt_batcher_open(os.fsencode(path), int(n), int(rec_bytes), int(threads))
        if not self._h:
            raise OSError(f"cannot map {path} as {n} records of {rec_bytes} bytes")
        self.n, self.rec_bytes, self.threads = int(n), int(rec_bytes), int(threads)
        self._keep = {}                      # ticket -> destination tensor: alive until its job is done

    def submit(self, indices: np.ndarray, dst: torch.Tensor) -> int:
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        if dst.dtype != torch.uint8 or not dst.is_contiguous() or dst.numel() < idx.size * self.rec_bytes:
            raise ValueError("gather destination must be a contiguous uint8 tensor of at least count * rec_bytes")
        t = self._lib.pdt_batcher_submit(self._h, idx.ctypes.data, int(idx.size), dst.data_ptr())
        if t <= 0:
            raise RuntimeError("pdt_batcher_submit failed")
        self._keep[int(t)] = dst
        return int(t)

    def wait(self, ticket: int) -> None:
        if not self._h:                          # closed: close() joined the threads, nothing is in flight any more
            self._keep.pop(int(ticket), None)
            return
        rc = self._lib.pdt_batcher_wait(self._h, int(ticket))
        self._keep.pop(int(ticket), None)
        if rc == 1:
            raise IndexError("a record index was out of range")
        if rc != 0:
            raise RuntimeError(f"pdt_batcher_wait failed ({rc})")

    def close(self) -> None:
        if self._h:
            self._lib.pdt_batcher_close(self._h)         # joins the threads: nothing writes after this
            self._h = None
        self._keep.clear()

    def __del__(self) -> None:       # noqa: D105
        try:
            self.close()
        except Exception:            # noqa: BLE001 -- interpreter shutdown
            pass


class NativeRecordLoader:
    """Iterable over ``(img [B, S, S, 3] uint8, label [B] int64)`` batches of a packed split, in the order of ``sampler``.

    ``depth`` batches are being gathered while the consumer works on the current one.  The batch tensors live in a ring of
    ``depth + 3`` buffers (pinned when a GPU is present): a buffer is written again only after the consumer has been handed
    two further batches (it may hold the current batch and the one before while asking for the next), and -- if the consumer copies it to the GPU asynchronously and says so through ``img.pdt_release(event)``
    (what :class:`..data.loader.DevicePrefetcher` does) -- only after that copy has completed."""

    def __init__(self, dataset, batch_size: int, sampler, threads: int, drop_last: bool = True, pin: bool = True,
                 depth: int = 2) -> None:
        path, n, shape, labels = dataset.native_records()
        self.dataset, self.sampler = dataset, sampler
        self.batch_size, self.drop_last = int(batch_size), bool(drop_last)
        self.shape = tuple(int(s) for s in shape)
        self.rec_bytes = int(np.prod(self.shape))
        self.labels = np.asarray(labels, dtype=np.int64)
        self.depth = max(1, int(depth))
        self.pin = bool(pin) and torch.cuda.is_available()
        self.threads = max(1, int(threads))
        self._gather = RecordGatherer(path, n, self.rec_bytes, self.threads)
        self._ring = None
        self._events = None

    def __len__(self) -> int:
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _buffers(self):
        if self._ring is None:
            k = self.depth + 3           # depth being gathered + the one handed out now + the two before it
            self._ring = [(torch.empty((self.batch_size, *self.shape), dtype=torch.uint8, pin_memory=self.pin),
                           torch.empty(self.batch_size, dtype=torch.int64, pin_memory=self.pin)) for _ in range(k)]
            self._events = [None] * k
        return self._ring

    def _index_batches(self) -> Iterator[np.ndarray]:
        cur = []
        for i in self.sampler:
            cur.append(int(i))
            if len(cur) == self.batch_size:
                yield np.asarray(cur, dtype=np.int64)
                cur = []
        if cur and not self.drop_last:
            yield np.asarray(cur, dtype=np.int64)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        ring = self._buffers()
        pending: deque = deque()
        slot = 0
        try:
            for idx in self._index_batches():
                ev = self._events[slot]
                if ev is not None:                       # the GPU copy out of this buffer, issued R batches ago
                    ev.synchronize()
                    self._events[slot] = None
                img, label = ring[slot]
                ticket = self._gather.submit(idx, img)
                label[:idx.size].copy_(torch.from_numpy(self.labels[idx]))
                pending.append((ticket, slot, idx.size))
                slot = (slot + 1) % len(ring)
                if len(pending) > self.depth:
                    yield self._finish(pending.popleft())
            while pending:
                yield self._finish(pending.popleft())
        finally:
            while pending:                               # abandoned mid-epoch: the threads must be done with the buffers
                self._gather.wait(pending.popleft()[0])

    def _finish(self, item) -> Tuple[torch.Tensor, torch.Tensor]:
        ticket, slot, count = item
        self._gather.wait(ticket)
        img, label = self._ring[slot]
        img, label = img[:count], label[:count]

        def release(event, _slot=slot) -> None:
            self._events[_slot] = event

        img.pdt_release = release
        return img, label

    def close(self) -> None:
        self._gather.close()
