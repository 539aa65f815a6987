"""TensorBoard event files without TensorBoard: ``events.out.tfevents.<ts>.<host>`` per trial.

A ``tune.run`` result tree carries one event file per trial next to ``result.json`` and ``progress.csv``
(the reference only names the directory, /root/reference/ray-tune-hpo-regression.py:476).  Neither
``tensorboard`` nor ``protobuf``-generated classes are needed for scalars: an event file is a sequence of
length-prefixed records, each protected by two masked CRC-32C values, and the ``Event`` / ``Summary``
messages used for scalars have five fields between them, so they are encoded by hand here.

    record  := u64le(len) u32le(mask(crc32c(len))) data u32le(mask(crc32c(data)))
    Event   := 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary
    Summary := repeated 1: Value ;  Value := 1: string tag, 2: float simple_value
"""
from __future__ import annotations

import os
import socket
import struct
import time
from typing import Dict, Iterator, List, Tuple

from .logger import Logger
from .space import flatten

_CRC_TABLE: List[int] = []


def _table() -> List[int]:
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1       # Castagnoli polynomial, reflected
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    t, c = _table(), 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _len_field(num: int, payload: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time: float, step: int, scalars: Dict[str, float] = None, file_version: str = None) -> bytes:
    ev = _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _varint((2 << 3) | 0) + _varint(step)
    if file_version is not None:
        ev += _len_field(3, file_version.encode())
    if scalars:
        summary = b""
        for tag, v in scalars.items():
            value = _len_field(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", v)
            summary += _len_field(1, value)
        ev += _len_field(5, summary)
    return ev


def frame(data: bytes) -> bytes:
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", _masked(head)) + data + struct.pack("<I", _masked(data))


def read_records(path: str) -> Iterator[bytes]:
    """Yield the payload of every record, verifying both checksums (used by the tests and by ``read_scalars``)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(8)
            if len(head) < 8:
                return
            (n,) = struct.unpack("<Q", head)
            (hc,) = struct.unpack("<I", f.read(4))
            data = f.read(n)
            (dc,) = struct.unpack("<I", f.read(4))
            if hc != _masked(head) or dc != _masked(data):
                raise ValueError(f"corrupt record in {path}")
            yield data


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    i = 0
    while i < len(buf):
        key = shift = 0
        while True:
            b = buf[i]; i += 1
            key |= (b & 0x7F) << shift; shift += 7
            if not b & 0x80:
                break
        num, wt = key >> 3, key & 7
        if wt == 0:
            v = shift = 0
            while True:
                b = buf[i]; i += 1
                v |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            yield num, wt, v
        elif wt == 1:
            yield num, wt, buf[i:i + 8]; i += 8
        elif wt == 5:
            yield num, wt, buf[i:i + 4]; i += 4
        elif wt == 2:
            n = shift = 0
            while True:
                b = buf[i]; i += 1
                n |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            yield num, wt, buf[i:i + n]; i += n
        else:
            raise ValueError(f"unsupported wire type {wt}")


def read_scalars(path: str) -> List[Tuple[int, str, float]]:
    out = []
    for rec in read_records(path):
        step, summary = 0, None
        for num, wt, v in _fields(rec):
            if num == 2 and wt == 0:
                step = v
            elif num == 5 and wt == 2:
                summary = v
        if summary is None:
            continue
        for num, wt, value in _fields(summary):
            tag, val = None, None
            for n2, w2, v2 in _fields(value):
                if n2 == 1:
                    tag = v2.decode()
                elif n2 == 2 and w2 == 5:
                    (val,) = struct.unpack("<f", v2)
            if tag is not None and val is not None:
                out.append((step, tag, val))
    return out


class TBXLogger(Logger):
    """Writes every numeric field of every result as a scalar under ``ray/tune/<key>`` at step ``training_iteration``."""

    PREFIX = "ray/tune/"

    def __init__(self):
        self._files = {}

    def _file(self, trial):
        f = self._files.get(trial.trial_id)
        if f is None:
            os.makedirs(trial.logdir, exist_ok=True)
            path = os.path.join(trial.logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}")
            f = open(path, "ab")
            f.write(frame(encode_event(time.time(), 0, file_version="brain.Event:2")))
            self._files[trial.trial_id] = f
        return f

    def on_result(self, trial, result):
        if not trial.logdir:
            return
        scalars = {}
        for k, v in flatten(result).items():
            if k.startswith("config") or isinstance(v, bool) or not isinstance(v, (int, float)):
                continue
            scalars[self.PREFIX + k] = float(v)
        f = self._file(trial)
        f.write(frame(encode_event(float(result.get("timestamp", time.time())), int(result.get("training_iteration", 0)),
                                   scalars)))
        f.flush()

    def on_trial_end(self, trial):
        f = self._files.pop(trial.trial_id, None)
        if f:
            f.close()

    def close(self):
        for f in self._files.values():
            f.close()
        self._files.clear()
