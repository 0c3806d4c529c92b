"""Scalar training logs in TensorBoard's event-file format, written and read without TensorBoard.

The reference hands `tensorboard_log` to SB3 (`train_rate.py:144`) and its plotting tools read the resulting
`events.out.tfevents.*` files back through `EventAccumulator` (`learned_controllers/visualize/learning_curves.py:35-84`:
every scalar tag -> rows of step / value / wall_time; tags `rollout/ep_rew_mean`, `train/value_loss`, ... :114-121).
TensorBoard is not in this image, so the file format is produced directly:

  file   = record*                       record = u64 length | u32 masked_crc32c(length) | data | u32 masked_crc32c(data)
  data   = Event protobuf                Event  = 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary
  Summary = repeated 1: Value            Value  = 1: string tag, 2: float simple_value
  masked_crc(x) = rotr15(crc32c(x)) + 0xa282ead8

(TFRecord framing and event.proto / summary.proto field numbers as published; the first record carries
file_version "brain.Event:2".)  `read_events` is the inverse, checks both CRCs of every record, and is what the tests
use -- parity with TensorBoard's own reader is unpinned (package absent).
"""
import os
import socket
import struct
import time
from typing import Dict, Iterator, List, Optional, Tuple

_POLY = 0x82F63B78          # CRC-32C (Castagnoli), reflected
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num: int, wire: int) -> bytes:
    return _varint((num << 3) | wire)


def _bytes_field(num: int, payload: bytes) -> bytes:
    return _field(num, 2) + _varint(len(payload)) + payload


def encode_event(wall_time: float, step: int, scalars: Optional[Dict[str, float]] = None, file_version: Optional[str] = None) -> bytes:
    ev = _field(1, 1) + struct.pack("<d", wall_time) + _field(2, 0) + _varint(int(step))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    if scalars:
        summary = b"".join(_bytes_field(1, _bytes_field(1, tag.encode()) + _field(2, 5) + struct.pack("<f", float(v)))
                           for tag, v in scalars.items())
        ev += _bytes_field(5, summary)
    return ev


def _record(data: bytes) -> bytes:
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data))


class EventFileWriter:
    """`add_scalar(tag, value, step)` / `add_scalars({...}, step)` -> `<log_dir>/events.out.tfevents.<time>.<host>.<pid>.0`."""

    def __init__(self, log_dir: str, filename_suffix: str = ""):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}.0{filename_suffix}")
        self._f = open(self.path, "ab")
        self._f.write(_record(encode_event(time.time(), 0, file_version="brain.Event:2")))
        self._f.flush()

    def add_scalars(self, scalars: Dict[str, float], step: int, wall_time: Optional[float] = None) -> None:
        self._f.write(_record(encode_event(time.time() if wall_time is None else wall_time, step, scalars)))

    def add_scalar(self, tag: str, value: float, step: int, wall_time: Optional[float] = None) -> None:
        self.add_scalars({tag: value}, step, wall_time)

    def flush(self) -> None:
        self._f.flush()

    def close(self) -> None:
        if not self._f.closed:
            self._f.close()


# ---- reader --------------------------------------------------------------------------------------------------------------
def _read_varint(buf: bytes, i: int) -> Tuple[int, int]:
    n, shift = 0, 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, i


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    i = 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        num, wire = key >> 3, key & 7
        if wire == 0:
            v, i = _read_varint(buf, i)
        elif wire == 1:
            v, i = buf[i:i + 8], i + 8
        elif wire == 5:
            v, i = buf[i:i + 4], i + 4
        elif wire == 2:
            n, i = _read_varint(buf, i)
            v, i = buf[i:i + n], i + n
        else:
            raise ValueError(f"unsupported wire type {wire}")
        yield num, wire, v


def read_events(path: str) -> List[dict]:
    """-> [{"wall_time", "step", "file_version" | None, "scalars": {tag: value}}, ...]; raises on a CRC mismatch."""
    out = []
    with open(path, "rb") as f:
        raw = f.read()
    i = 0
    while i < len(raw):
        head = raw[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", raw[i + 8:i + 12])[0] != masked_crc32c(head):
            raise ValueError(f"{path}: length CRC mismatch at byte {i}")
        data = raw[i + 12:i + 12 + n]
        if struct.unpack("<I", raw[i + 12 + n:i + 16 + n])[0] != masked_crc32c(data):
            raise ValueError(f"{path}: data CRC mismatch at byte {i}")
        i += 16 + n
        ev = {"wall_time": 0.0, "step": 0, "file_version": None, "scalars": {}}
        for num, _w, v in _fields(data):
            if num == 1:
                ev["wall_time"] = struct.unpack("<d", v)[0]
            elif num == 2:
                ev["step"] = v
            elif num == 3:
                ev["file_version"] = v.decode()
            elif num == 5:
                for n2, _w2, val in _fields(v):
                    if n2 != 1:
                        continue
                    tag, simple = None, None
                    for n3, _w3, x in _fields(val):
                        if n3 == 1:
                            tag = x.decode()
                        elif n3 == 2:
                            simple = struct.unpack("<f", x)[0]
                    if tag is not None and simple is not None:
                        ev["scalars"][tag] = simple
        out.append(ev)
    return out


def load_scalars(log_dir: str) -> Dict[str, List[Tuple[int, float, float]]]:
    """Every `events.out.tfevents.*` under `log_dir` -> {tag: [(step, value, wall_time), ...] sorted by step}: the table
    learning_curves.py:35-84 builds."""
    table: Dict[str, List[Tuple[int, float, float]]] = {}
    for root, _dirs, files in os.walk(log_dir):
        for name in sorted(files):
            if name.startswith("events.out.tfevents."):
                for ev in read_events(os.path.join(root, name)):
                    for tag, v in ev["scalars"].items():
                        table.setdefault(tag, []).append((ev["step"], v, ev["wall_time"]))
    for rows in table.values():
        rows.sort(key=lambda r: r[0])
    return table
