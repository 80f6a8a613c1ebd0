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
always appends to ``<log_dir>/tf-board-logs/<name>/metrics.jsonl`` AND writes a TensorBoard
event file ``events.out.tfevents.*`` next to it with :class:`EventFileWriter`, an in-tree
writer of the TFRecord / ``tensorflow.Event`` wire format (the ``tensorboard`` package is not
needed to produce the file, only to look at it).
"""
from __future__ import annotations

import json
import logging
import logging.handlers
import multiprocessing as mp
import os
import socket
import struct
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


def _crc32c_table():
    table = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    return table


_CRC32C = _crc32c_table()


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli polynomial, reflected), the checksum of the TFRecord framing."""
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC32C[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= 0xFFFFFFFFFFFFFFFF           # int64 fields: negative values take ten bytes, as protobuf encodes them
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _len_delimited(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_scalar_event(tag: str, value: float, step: int, wall_time: float) -> bytes:
    """``tensorflow.Event{wall_time = 1, step = 2, summary = 5{value = 1{tag = 1, simple_value = 2}}}``."""
    val = _len_delimited(1, tag.encode("utf-8")) + b"\x15" + struct.pack("<f", value)
    summary = _len_delimited(1, val)
    return b"\x09" + struct.pack("<d", wall_time) + b"\x10" + _varint(int(step)) + _len_delimited(5, summary)


def encode_version_event(wall_time: float) -> bytes:
    """The first record of every event file: ``Event{wall_time = 1, file_version = 3}``."""
    return b"\x09" + struct.pack("<d", wall_time) + _len_delimited(3, b"brain.Event:2")


def frame_record(data: bytes) -> bytes:
    """TFRecord framing: length (u64 LE), masked CRC of the length, payload, masked CRC of the payload."""
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data))


class EventFileWriter:
    """TensorBoard scalar event file, written without the ``tensorboard`` package: what upstream's
    ``SummaryWriter.add_scalar`` calls leave on disk (reference ``train_distributed.py:23,295-297,329-331``)."""

    _seq = 0

    def __init__(self, log_dir: str) -> None:
        os.makedirs(log_dir, exist_ok=True)
        EventFileWriter._seq += 1
        name = f"events.out.tfevents.{int(time.time()):010d}.{socket.gethostname()}.{os.getpid()}.{EventFileWriter._seq}"
        self.path = os.path.join(log_dir, name)
        self._f = open(self.path, "ab")
        self._f.write(frame_record(encode_version_event(time.time())))
        self._f.flush()

    def add_scalar(self, tag: str, value: float, step: int, wall_time: Optional[float] = None) -> None:
        wall = time.time() if wall_time is None else wall_time
        self._f.write(frame_record(encode_scalar_event(tag, float(value), int(step), wall)))

    def flush(self) -> None:
        self._f.flush()

    def close(self) -> None:
        if not self._f.closed:
            self._f.flush()
            self._f.close()


def read_event_file(path: str):
    """Check the framing of an event file and decode it: ``[(wall_time, step, tag, value)]`` for the scalar
    records, ``(wall_time, None, "file_version", text)`` for the header.  Raises ``ValueError`` on a bad checksum."""

    def read_varint(buf: bytes, pos: int):
        shift = n = 0
        while True:
            b = buf[pos]
            pos += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, pos

    def fields(buf: bytes):
        pos = 0
        while pos < len(buf):
            key, pos = read_varint(buf, pos)
            field, wire = key >> 3, key & 7
            if wire == 0:
                v, pos = read_varint(buf, pos)
            elif wire == 1:
                v, pos = buf[pos:pos + 8], pos + 8
            elif wire == 5:
                v, pos = buf[pos:pos + 4], pos + 4
            elif wire == 2:
                n, pos = read_varint(buf, pos)
                v, pos = buf[pos:pos + n], pos + n
            else:
                raise ValueError(f"unsupported wire type {wire}")
            yield field, wire, v

    out = []
    with open(path, "rb") as f:
        blob = f.read()
    pos = 0
    while pos < len(blob):
        head = blob[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        (hcrc,) = struct.unpack("<I", blob[pos + 8:pos + 12])
        data = blob[pos + 12:pos + 12 + n]
        (dcrc,) = struct.unpack("<I", blob[pos + 12 + n:pos + 16 + n])
        if hcrc != masked_crc32c(head) or dcrc != masked_crc32c(data) or len(data) != n:
            raise ValueError(f"{path}: corrupt record at byte {pos}")
        pos += 16 + n
        wall, step, version, scalars = 0.0, 0, None, []
        for field, wire, v in fields(data):
            if field == 1 and wire == 1:
                (wall,) = struct.unpack("<d", v)
            elif field == 2 and wire == 0:
                step = v - (1 << 64) if v >> 63 else v
            elif field == 3 and wire == 2:
                version = v.decode()
            elif field == 5 and wire == 2:
                for f1, w1, value_msg in fields(v):
                    if f1 != 1 or w1 != 2:
                        continue
                    tag, val = None, None
                    for f2, w2, x in fields(value_msg):
                        if f2 == 1 and w2 == 2:
                            tag = x.decode()
                        elif f2 == 2 and w2 == 5:
                            (val,) = struct.unpack("<f", x)
                    scalars.append((tag, val))
        if version is not None:
            out.append((wall, None, "file_version", version))
        for tag, val in scalars:
            out.append((wall, step, tag, val))
    return out


class ScalarWriter:
    """``add_scalar(tag, value, step)`` -> ``metrics.jsonl`` and a TensorBoard event file."""

    def __init__(self, log_dir: str) -> None:
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self._f = open(os.path.join(log_dir, "metrics.jsonl"), "a", buffering=1)
        self._tb = EventFileWriter(log_dir)

    def add_scalar(self, tag: str, value, step: int) -> None:
        value = float(value)
        now = time.time()
        self._f.write(json.dumps({"tag": tag, "value": value, "step": int(step), "time": now}) + "\n")
        self._tb.add_scalar(tag, value, step, wall_time=now)

    def add_scalars(self, values: dict, step: int) -> None:
        for tag, v in values.items():
            self.add_scalar(tag, v, step)

    def flush(self) -> None:
        self._f.flush()
        self._tb.flush()

    def close(self) -> None:
        self.flush()
        self._f.close()
        self._tb.close()


def get_tb_writer(log_dir: Optional[str], filename: Optional[str]) -> ScalarWriter:
    """Scalar writer under ``<log_dir>/tf-board-logs/<filename>``."""
    base = os.path.join(log_dir or ".", "tf-board-logs", filename or "run")
    return ScalarWriter(base)
