"""TensorBoard event files written without the ``tensorboard`` package (upstream writes its scalars with
``SummaryWriter``, reference ``train_distributed.py:23,295-297,329-331``): checksum against the published CRC-32C test
vectors, the record bytes of a scalar event against a hand-assembled golden, and a writer -> reader round trip."""
import glob
import os
import struct

import pytest

from pytorch_distributed_training_amd import logging_utils as lu


def test_crc32c_known_vectors():
    # RFC 3720 (iSCSI) appendix B.4 and the standard check value
    assert lu.crc32c(b"123456789") == 0xE3069283
    assert lu.crc32c(bytes(32)) == 0x8A9136AA
    assert lu.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert lu.crc32c(bytes(range(32))) == 0x46DD794E
    # TFRecord masking: rotate right by 15 and add a constant
    c = 0xE3069283
    assert lu.masked_crc32c(b"123456789") == ((((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF) == 0xC78AB0E5


def test_scalar_event_golden_bytes():
    # Event{wall_time(1, fixed64) = 1.0, step(2, varint) = 7, summary(5){value(1){tag(1) = "a", simple_value(2, fixed32) = 0.5}}}
    value = bytes([0x0A, 0x01]) + b"a" + bytes([0x15]) + struct.pack("<f", 0.5)
    summary = bytes([0x0A, len(value)]) + value
    event = bytes([0x09]) + struct.pack("<d", 1.0) + bytes([0x10, 0x07]) + bytes([0x2A, len(summary)]) + summary
    assert lu.encode_scalar_event("a", 0.5, 7, 1.0) == event
    rec = lu.frame_record(event)
    assert rec.hex() == "1700000000000000e7cef81e09000000000000f03f10072a0a0a080a0161150000003f35e3065a"
    assert len(rec) == 8 + 4 + len(event) + 4 and struct.unpack("<Q", rec[:8])[0] == len(event)
    # a step above 127 takes a two-byte varint, a negative one ten bytes (int64 two's complement)
    assert lu.encode_scalar_event("a", 0.5, 300, 1.0)[9:12] == bytes([0x10, 0xAC, 0x02])
    assert len(lu._varint(-1)) == 10
    # the file header every reader expects first
    head = lu.encode_version_event(2.0)
    assert head == bytes([0x09]) + struct.pack("<d", 2.0) + bytes([0x1A, 13]) + b"brain.Event:2"


def test_scalar_writer_round_trip(tmp_path):
    w = lu.get_tb_writer(str(tmp_path), "run1")
    w.add_scalar("loss/train", 1.5, 0)
    w.add_scalars({"eval/Acc@1": 71.25, "eval/loss": 0.125}, 2000)
    w.close()
    base = os.path.join(tmp_path, "tf-board-logs", "run1")
    files = glob.glob(os.path.join(base, "events.out.tfevents.*"))
    assert len(files) == 1 and os.path.exists(os.path.join(base, "metrics.jsonl"))
    ev = lu.read_event_file(files[0])
    assert ev[0][2:] == ("file_version", "brain.Event:2")
    assert [(s, t, v) for _, s, t, v in ev[1:]] == [(0, "loss/train", 1.5), (2000, "eval/Acc@1", 71.25), (2000, "eval/loss", 0.125)]
    # a flipped payload byte is caught by the record checksum
    blob = bytearray(open(files[0], "rb").read())
    blob[-6] ^= 0x01
    bad = os.path.join(tmp_path, "bad")
    open(bad, "wb").write(bytes(blob))
    with pytest.raises(ValueError):
        lu.read_event_file(bad)
