"""TF-free reader of TensorFlow "tensor bundle" checkpoints - the `variables/variables.index` +
`variables/variables.data-0000N-of-0000M` pair inside a Keras SavedModel directory.

This is what replaces the variable-restore half of ``tf.compat.v2.saved_model.load(model_path)``
(reference eval/interpolator.py:148; the files are written by ``model.save()``, training/train_lib.py:280 and
training/build_saved_model_cli.py:65-73).  TensorFlow is not needed: the formats are restated here from their
published definitions.

  * ``.index`` is a LevelDB-style sorted string table (tensorflow/core/lib/io/table_format.txt):
        [data block]* [metaindex block] [index block] [48-byte footer]
    footer  = BlockHandle(metaindex) BlockHandle(index), zero padding to 40 bytes, magic 0xdb4775248b80fb57 (LE).
    block   = entries, uint32 restart offsets, uint32 restart count; then 1 type byte (0 raw, 1 snappy) and a
              4-byte masked crc32c over (contents + type).  BlockHandle = varint64 offset, varint64 size
              (size excludes the 5-byte trailer).
    entry   = varint32 shared key bytes, varint32 unshared key bytes, varint32 value length, key suffix, value.
    The index block maps a separator key (>= last key of a data block) to that block's handle.
  * values are protobufs (tensorflow/core/protobuf/tensor_bundle.proto): key "" -> BundleHeaderProto
    {num_shards=1, endianness=2, version=3}; every other key -> BundleEntryProto {dtype=1, shape=2, shard_id=3,
    offset=4, size=5, crc32c=6 (fixed32, masked), slices=7}.  dtype DT_FLOAT = 1.  Tensor bytes are raw
    row-major little-endian at [offset, offset+size) of shard `shard_id`.
  * Keras object-graph checkpoint keys are the attribute path from the root object plus
    "/.ATTRIBUTES/VARIABLE_VALUE".  For film_net the paths follow the attribute names in the reference source:
        .../extract_sublevels/convs/{i}/{kernel,bias}        feature_extractor.py:118-123,160
        .../_predictors/{p}/_convs/{j}/{kernel,bias}          pyramid_flow_estimator.py:74-83,111-123
        .../convs/{i}/{j}/{kernel,bias}, .../output_conv/...  fusion.py:64-101
    (UNVERIFIED against a real published checkpoint: none is reachable from this environment.  The mapping is
    therefore pattern based, independent of the `layer_with_weights-N` numbering, and falls back to matching by
    tensor shape only where the shape is unique; `load_film_weights` reports which rule placed every tensor and logs a
    warning whenever one was placed by shape.  tests/tf_like_writer.py is a second, independently written writer that
    lays a bundle out the way TensorFlow's writers do - several shards, multi-block index with shortest separators,
    Adam slot variables, int64 counters, a real TrackableObjectGraph - and the reader is tested against it.)

crc32c (Castagnoli) of large shards is computed by the native helper in libfilm_hip.so (film_crc32c) when the
library is built; a table-driven pure-Python fallback covers the (small) index blocks.

Round 5: the PRODUCT path no longer parses bundles in Python - `Interpolator(<SavedModel dir>)` calls the native reader behind the
C-ABI (film_load_bundle, csrc/film_bundle.cpp: same format statement, same placement rules).  This module stays as the second,
independent reader the tests hold the native one against (tests/test_tf_bundle_cpu.py), as the inspection tool
(BundleReader.keys / object_graph_keys) and behind `film_hip.weights.load_weights` for callers that want the arrays themselves.
The bundle WRITERS used to validate both readers live with the tests (tests/bundle_writer.py, tests/tf_like_writer.py).
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_TRAILER = 5
DT_FLOAT = 1
DT_STRING = 7
VAR_SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'


# ------------------------------------------------------------------------------------------------ crc32c
def _make_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC_TABLE = _make_table()


def _crc32c_py(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data, crc: int = 0) -> int:
    """CRC-32C of a bytes-like object / contiguous numpy array (native when libfilm_hip.so is available)."""
    mv = memoryview(data).cast('B')
    if len(mv) >= 4096:
        try:
            from . import engine
            lib = engine.load_library()
            arr = np.frombuffer(mv, dtype=np.uint8)
            return int(lib.film_crc32c(crc, arr.ctypes.data, len(mv)))
        except Exception:
            pass
    return _crc32c_py(bytes(mv), crc)


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def unmask_crc(masked: int) -> int:
    rot = (masked - 0xa282ead8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ varints / protobuf
def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise ValueError('truncated varint')
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError('varint too long')


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def pb_decode(buf: bytes) -> List[Tuple[int, int, object]]:
    """Minimal protobuf wire decoder: [(field number, wire type, value)]; length-delimited values stay bytes."""
    out = []
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise ValueError('truncated protobuf field')
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        out.append((field, wt, v))
    return out


def _pb_field(field: int, wt: int, payload: bytes) -> bytes:
    return _put_varint((field << 3) | wt) + payload


def _pb_varint(field: int, v: int) -> bytes:
    return _pb_field(field, 0, _put_varint(v & 0xFFFFFFFFFFFFFFFF))


def _pb_bytes(field: int, v: bytes) -> bytes:
    return _pb_field(field, 2, _put_varint(len(v)) + v)


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


# ------------------------------------------------------------------------------------------------ table reader
def _read_block(data: bytes, offset: int, size: int, verify: bool) -> bytes:
    if offset + size + BLOCK_TRAILER > len(data):
        raise ValueError('block handle points outside the index file')
    contents = data[offset:offset + size]
    ctype = data[offset + size]
    stored = struct.unpack_from('<I', data, offset + size + 1)[0]
    if verify and unmask_crc(stored) != _crc32c_py(data[offset:offset + size + 1]):
        raise ValueError(f'index block at {offset}: crc32c mismatch')
    if ctype == 1:
        raise ValueError('snappy-compressed index blocks are not supported (TensorFlow writes bundles uncompressed)')
    if ctype != 0:
        raise ValueError(f'unknown block compression type {ctype}')
    return contents


def _block_entries(block: bytes) -> Iterable[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise ValueError('block too small')
    nrestarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * nrestarts
    if limit < 0:
        raise ValueError('bad restart array')
    pos = 0
    key = b''
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + unshared + vlen > limit:
            raise ValueError('corrupt block entry')
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path: str, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < FOOTER_LEN:
        raise ValueError(f'{path}: too small to be a table file')
    footer = data[-FOOTER_LEN:]
    if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
        raise ValueError(f'{path}: bad table magic (not a TensorFlow bundle index)')
    pos = 0
    _mo, pos = _get_varint(footer, pos)
    _ms, pos = _get_varint(footer, pos)
    io_, pos = _get_varint(footer, pos)
    is_, pos = _get_varint(footer, pos)
    out = []
    for _sep, handle in _block_entries(_read_block(data, io_, is_, verify)):
        bo, p = _get_varint(handle, 0)
        bs, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(data, bo, bs, verify)))
    return out


# ------------------------------------------------------------------------------------------------ bundle reader
class BundleEntry:
    def __init__(self, dtype=0, shape=(), shard_id=0, offset=0, size=0, crc=0, has_slices=False):
        self.dtype, self.shape, self.shard_id = dtype, tuple(shape), shard_id
        self.offset, self.size, self.crc, self.has_slices = offset, size, crc, has_slices

    def __repr__(self):
        return f'BundleEntry(dtype={self.dtype}, shape={self.shape}, shard={self.shard_id}, offset={self.offset}, size={self.size})'


def _parse_entry(value: bytes) -> BundleEntry:
    e = BundleEntry()
    for field, _wt, v in pb_decode(value):
        if field == 1:
            e.dtype = v
        elif field == 2:
            dims = []
            for f2, _w2, v2 in pb_decode(v):
                if f2 == 2:
                    size = 0
                    for f3, _w3, v3 in pb_decode(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    dims.append(size)
            e.shape = tuple(dims)
        elif field == 3:
            e.shard_id = v
        elif field == 4:
            e.offset = v
        elif field == 5:
            e.size = v
        elif field == 6:
            e.crc = v
        elif field == 7:
            e.has_slices = True
    return e


class BundleReader:
    """Reads `<prefix>.index` + `<prefix>.data-0000N-of-0000M`."""

    def __init__(self, prefix: str, verify: bool = True):
        self.prefix = prefix
        self.verify = verify
        entries = read_table(prefix + '.index', verify)
        if not entries or entries[0][0] != b'':
            raise ValueError(f'{prefix}.index: missing bundle header entry')
        self.num_shards, endianness = 1, 0
        for field, _wt, v in pb_decode(entries[0][1]):
            if field == 1:
                self.num_shards = v
            elif field == 2:
                endianness = v
        if endianness != 0:
            raise ValueError('big-endian bundles are not supported')
        self.entries: Dict[str, BundleEntry] = {k.decode('utf-8'): _parse_entry(v) for k, v in entries[1:]}
        self._shards: Dict[int, np.memmap] = {}

    def keys(self) -> List[str]:
        return list(self.entries)

    def _shard(self, i: int):
        if i not in self._shards:
            fn = f'{self.prefix}.data-{i:05d}-of-{self.num_shards:05d}'
            self._shards[i] = np.memmap(fn, dtype=np.uint8, mode='r')
        return self._shards[i]

    def raw(self, key: str) -> bytes:
        e = self.entries[key]
        if e.has_slices:
            raise ValueError(f'{key}: sliced (partitioned) variables are not supported')
        sh = self._shard(e.shard_id)
        if e.offset + e.size > sh.shape[0]:
            raise ValueError(f'{key}: data range outside shard {e.shard_id}')
        buf = np.asarray(sh[e.offset:e.offset + e.size])
        if self.verify and unmask_crc(e.crc) != crc32c(buf):
            raise ValueError(f'{key}: tensor crc32c mismatch')
        return buf.tobytes()

    def tensor(self, key: str) -> np.ndarray:
        e = self.entries[key]
        if e.dtype != DT_FLOAT:
            raise ValueError(f'{key}: dtype {e.dtype} is not DT_FLOAT')
        n = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        if e.size != 4 * n:
            raise ValueError(f'{key}: {e.size} bytes for shape {e.shape}')
        return np.frombuffer(self.raw(key), dtype='<f4').reshape(e.shape).copy()

    def string_scalar(self, key: str) -> bytes:
        """Scalar DT_STRING tensor: varint64 length, 4-byte masked crc32c of the length bytes, payload."""
        e = self.entries[key]
        if e.dtype != DT_STRING:
            raise ValueError(f'{key}: not a string tensor')
        sh = self._shard(e.shard_id)
        buf = bytes(np.asarray(sh[e.offset:e.offset + e.size]))
        n, pos = _get_varint(buf, 0)
        return buf[pos + 4:pos + 4 + n]

    def object_graph_keys(self) -> Dict[str, str]:
        """checkpoint_key -> full variable name, from the TrackableObjectGraph (trackable_object_graph.proto:
        nodes=1 {children=1, attributes=2 {name=1, full_name=2, checkpoint_key=3}})."""
        if OBJECT_GRAPH_KEY not in self.entries:
            return {}
        out = {}
        for f, _w, node in pb_decode(self.string_scalar(OBJECT_GRAPH_KEY)):
            if f != 1:
                continue
            for f2, _w2, attr in pb_decode(node):
                if f2 != 2:
                    continue
                full = ck = ''
                for f3, _w3, v3 in pb_decode(attr):
                    if f3 == 2:
                        full = v3.decode('utf-8')
                    elif f3 == 3:
                        ck = v3.decode('utf-8')
                if ck:
                    out[ck] = full
        return out


# ------------------------------------------------------------------------------------------------ film_net key mapping
_RE_FEAT = re.compile(r'(?:^|/)extract_sublevels/convs/([0-9]+)/(kernel|bias)$')
_RE_FLOW = re.compile(r'(?:^|/)_predictors/([0-9]+)/_convs/([0-9]+)/(kernel|bias)$')
_RE_FUSE = re.compile(r'(?:^|/)convs/([0-9]+)/([0-9]+)/(kernel|bias)$')
_RE_OUT = re.compile(r'(?:^|/)output_conv/(kernel|bias)$')


MAX_KEY_LEN = 4096      # longer keys, and numeric components of more than 9 digits, are not film_net weights (the native reader's limits)


def canonical_name(path: str, specialized_levels: int) -> Optional[str]:
    """Object-graph attribute path (without the VARIABLE_VALUE suffix) -> canonical tensor name, or None."""
    if len(path) > MAX_KEY_LEN or any(len(c) > 9 and c.isdigit() for c in path.rsplit('/', 5)[1:]):
        return None
    m = _RE_FEAT.search(path)
    if m:
        return f'feat_net/sub_extractor/cfeat_conv_{int(m.group(1))}/{m.group(2)}'
    m = _RE_FLOW.search(path)
    if m:
        p = int(m.group(1))
        pred = f'flow_predictor_{p}' if p < specialized_levels else 'flow_predictor_shared'
        return f'predict_flow/{pred}/conv_{int(m.group(2))}/{m.group(3)}'
    m = _RE_FUSE.search(path)
    if m:
        return f'fusion/convs_{int(m.group(1))}_{int(m.group(2))}/{m.group(3)}'
    m = _RE_OUT.search(path)
    if m:
        return f'fusion/output_conv/{m.group(1)}'
    return None


def checkpoint_key(name: str, opt) -> str:
    """Canonical tensor name -> the object-graph checkpoint key `model.save()` is expected to write
    (create_model wires feat_net, predict_flow, fusion in this order: interpolator.py:131-133,140,186)."""
    layer, rest = name.split('/', 1)
    leaf = rest.rsplit('/', 1)[1]
    if layer == 'feat_net':
        i = int(re.search(r'cfeat_conv_(\d+)', rest).group(1))
        path = f'layer_with_weights-0/extract_sublevels/convs/{i}/{leaf}'
    elif layer == 'predict_flow':
        m = re.search(r'flow_predictor_(\w+)/conv_(\d+)', rest)
        p = opt.specialized_levels if m.group(1) == 'shared' else int(m.group(1))
        path = f'layer_with_weights-1/_predictors/{p}/_convs/{int(m.group(2))}/{leaf}'
    elif rest.startswith('output_conv'):
        path = f'layer_with_weights-2/output_conv/{leaf}'
    else:
        m = re.search(r'convs_(\d+)_(\d+)', rest)
        path = f'layer_with_weights-2/convs/{int(m.group(1))}/{int(m.group(2))}/{leaf}'
    return path + VAR_SUFFIX


def _natural(s: str):
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', s)]


def load_film_weights(prefix: str, opt=None, verify: bool = True, report: Optional[dict] = None) -> Dict[str, np.ndarray]:
    """Reads every film_net tensor of a bundle into {canonical name: float32 array}.

    Rule 1: attribute-path patterns (see module docstring; the paths are the ones the reference's own layer code
    produces - tests/test_ref_golden_cpu.py derives them by running it).  Rule 2, for whatever rule 1 did not place:
    the unused float variable of the required shape, ONLY if that shape is unique on both sides; anything ambiguous
    raises instead of guessing.  `report[name]` = (rule, key)."""
    from . import weights as W
    from .options import PUBLISHED
    opt = opt or PUBLISHED
    rd = BundleReader(prefix, verify)
    specs = {}
    for name, shape, _act in W.weight_specs(opt):
        specs[name + '/kernel'] = tuple(shape)
        specs[name + '/bias'] = (shape[3],)
    out: Dict[str, np.ndarray] = {}
    used = set()
    rep = report if report is not None else {}
    var_keys = [k for k in rd.keys() if k.endswith(VAR_SUFFIX) and rd.entries[k].dtype == DT_FLOAT
                and '/.OPTIMIZER_SLOT/' not in k and not k.startswith('optimizer')]
    for k in var_keys:
        name = canonical_name(k[:-len(VAR_SUFFIX)], opt.specialized_levels)
        if name in specs and rd.entries[k].shape == specs[name]:
            t = rd.tensor(k)
            if name in out and not np.array_equal(out[name], t):
                raise ValueError(f'{name}: two different variables map to it ({rep[name][1]} and {k})')
            out[name] = t
            rep[name] = ('path', k)
            used.add(k)
    missing = [n for n in specs if n not in out]
    if missing:
        # Rule 2 places a tensor ONLY when the match is unambiguous: exactly one unused variable has the shape and
        # exactly one unplaced tensor wants it.  (Shapes repeat inside film_net - (3,3,256,256) is cfeat_conv_5,
        # flow_predictor_shared/conv_1,2 and fusion/convs_2_2 - so "first fit" could silently permute weights.)
        pool = sorted((k for k in var_keys if k not in used), key=_natural)
        ambiguous = []
        for name in missing:
            cands = [k for k in pool if k not in used and rd.entries[k].shape == specs[name]]
            rivals = [n for n in missing if n not in out and specs[n] == specs[name]]
            if len(cands) == 1 and len(rivals) == 1:
                out[name] = rd.tensor(cands[0])
                rep[name] = ('shape', cands[0])
                used.add(cands[0])
            elif cands:
                ambiguous.append((name, cands[:4]))
        if ambiguous:
            lines = '; '.join(f'{n} <- one of {c}' for n, c in ambiguous[:6])
            raise ValueError(f'{prefix}: {len(ambiguous)} tensor(s) could not be placed by their object-graph path and '
                             f'their shape is not unique among the remaining variables - refusing to guess: {lines}')
    still = [n for n in specs if n not in out]
    if still:
        raise KeyError(f'{prefix}: no variable found for {still[:4]}{"..." if len(still) > 4 else ""} '
                       f'({len(still)} of {len(specs)} tensors); keys look like {var_keys[:3]}')
    by_shape = sorted(n for n, (rule, _) in rep.items() if rule != 'path')
    if by_shape:   # say so by default: a tensor placed by its shape is a (unique-shape) guess, not a name match
        import logging
        logging.getLogger('film_hip.tf_bundle').warning(
            '%s: %d of %d tensors were not found under a known object-graph path and were placed by their (unique) shape: %s',
            prefix, len(by_shape), len(specs), ', '.join(f'{n} <- {rep[n][1]}' for n in by_shape))
    return out
