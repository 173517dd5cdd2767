"""Writer of TensorFlow "tensor bundle" checkpoints in the layout a Keras `model.save()` of film_net is expected to produce -
TEST INFRASTRUCTURE (moved out of the product module film_hip/tf_bundle.py in round 5): it exists to give the two bundle
readers (film_hip/tf_bundle.py in Python, csrc/film_bundle.cpp behind film_load_bundle) something to read, next to the second,
independently written tests/tf_like_writer.py.  Format: see the header of film_hip/tf_bundle.py."""
import os
import struct
from typing import Dict, List, Tuple

import numpy as np

from film_hip.tf_bundle import (DT_FLOAT, DT_STRING, OBJECT_GRAPH_KEY, TABLE_MAGIC, VAR_SUFFIX, _crc32c_py, _pb_bytes, _pb_field,
                                _pb_varint, _put_varint, checkpoint_key, crc32c, mask_crc)


class _BlockBuilder:
    def __init__(self, restart_interval: int = 16):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''
        self.interval = restart_interval

    def add(self, key: bytes, value: bytes) -> None:
        shared = 0
        if self.count < self.interval:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self) -> bytes:
        out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))
        return out

    def size(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4


def write_table(path: str, items: List[Tuple[bytes, bytes]], block_size: int = 4096) -> None:
    """Writes sorted (key, value) pairs as a LevelDB-format table (uncompressed blocks, as TensorFlow does)."""
    items = sorted(items, key=lambda kv: kv[0])
    out = bytearray()
    index = _BlockBuilder(restart_interval=1)

    def emit(contents: bytes) -> bytes:
        off = len(out)
        out.extend(contents)
        out.append(0)
        out.extend(struct.pack('<I', mask_crc(_crc32c_py(contents + b'\x00'))))
        return _put_varint(off) + _put_varint(len(contents))

    blk = _BlockBuilder()
    last_key = b''
    for k, v in items:
        blk.add(k, v)
        last_key = k
        if blk.size() >= block_size:
            index.add(last_key, emit(blk.finish()))
            blk = _BlockBuilder()
    if blk.count or not items:
        index.add(last_key, emit(blk.finish()))
    meta_handle = emit(_BlockBuilder().finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out.extend(footer)
    with open(path, 'wb') as f:
        f.write(out)


def _entry_proto(dtype: int, shape, offset: int, size: int, crc: int) -> bytes:
    dims = b''.join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)
    e = _pb_varint(1, dtype) + _pb_bytes(2, dims)
    if offset:
        e += _pb_varint(4, offset)
    e += _pb_varint(5, size) + _pb_field(6, 5, struct.pack('<I', mask_crc(crc)))
    return e


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], object_graph: bool = True) -> None:
    """Writes {checkpoint key: float32 array} as a one-shard bundle, plus (optionally) a
    _CHECKPOINTABLE_OBJECT_GRAPH string entry listing the keys - enough structure for the reader's tests and for
    exporting synthetic weights in the layout `Interpolator(model_path)` accepts."""
    os.makedirs(os.path.dirname(prefix) or '.', exist_ok=True)
    data = bytearray()
    items = [(b'', _pb_varint(1, 1) + _pb_varint(2, 0) + _pb_bytes(3, _pb_varint(1, 1)))]
    for key in sorted(tensors):
        arr = np.ascontiguousarray(tensors[key], dtype='<f4')
        raw = arr.tobytes()
        items.append((key.encode('utf-8'), _entry_proto(DT_FLOAT, arr.shape, len(data), len(raw), crc32c(raw))))
        data += raw
    if object_graph:
        nodes = b''
        for key in sorted(tensors):
            attr = _pb_bytes(1, b'VARIABLE_VALUE') + _pb_bytes(2, key[:-len(VAR_SUFFIX)].encode()) + _pb_bytes(3, key.encode())
            nodes += _pb_bytes(1, _pb_bytes(2, attr))
        lens = _put_varint(len(nodes))
        raw = lens + struct.pack('<I', mask_crc(_crc32c_py(lens))) + nodes
        items.append((OBJECT_GRAPH_KEY.encode(), _entry_proto(DT_STRING, (), len(data), len(raw), crc32c(raw))))
        data += raw
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(data)
    write_table(prefix + '.index', items)


def save_film_bundle(model_path: str, weights: Dict[str, np.ndarray], opt=None) -> str:
    """Writes `<model_path>/variables/variables.{index,data-00000-of-00001}` with the object-graph keys a Keras
    `model.save()` of film_net is expected to use; returns the bundle prefix."""
    from film_hip.options import PUBLISHED
    opt = opt or PUBLISHED
    prefix = os.path.join(model_path, 'variables', 'variables')
    write_bundle(prefix, {checkpoint_key(n, opt): w for n, w in weights.items()})
    return prefix
