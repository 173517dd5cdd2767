"""A SECOND, independently written tensor-bundle writer that lays a checkpoint out the way TensorFlow's own writers do -
test infrastructure for film_hip/tf_bundle.py's reader, which otherwise only ever met the output of its own writer.

It shares no code with film_hip/tf_bundle.py (own varint / protobuf / crc32c / block builder) and deliberately differs from
that module's writer wherever TensorFlow does:

  * SEVERAL data shards (`variables.data-0000i-of-0000N`, BundleEntryProto.shard_id), as MergeBundles leaves them;
  * a multi-block index (small block_size) whose data blocks use LevelDB's restart interval of 16 with prefix compression
    between restart points, and whose INDEX block holds shortest separators (BytewiseComparator::FindShortestSeparator),
    not the last key of each block;
  * everything a Keras `model.save()` / `tf.train.Checkpoint` puts next to the model variables: Adam slot variables under
    `<variable path>/.OPTIMIZER_SLOT/optimizer/{m,v}/.ATTRIBUTES/VARIABLE_VALUE` (same shapes as the variables - a trap
    for shape matching), `optimizer/iter` and `save_counter` (DT_INT64 scalars), float hyper-parameter scalars;
  * the `_CHECKPOINTABLE_OBJECT_GRAPH` entry as a real TrackableObjectGraph message (trackable_object_graph.proto): a node
    per Python object with `children {node_id, local_name}` edges from the root down to every variable, `attributes
    {name, full_name, checkpoint_key}` on the variable nodes and `slot_variables` on the optimizer node - stored as a
    DT_STRING scalar (varint64 length, masked crc32c of the length bytes, payload);
  * padding bytes between tensors, so offsets - not running sums - must be used.

Formats: tensorflow/core/lib/io/table_format.txt, tensorflow/core/protobuf/tensor_bundle.proto,
tensorflow/core/protobuf/trackable_object_graph.proto, tensorflow/core/util/tensor_bundle/tensor_bundle.cc (string tensors).
"""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_STRING, DT_INT64 = 1, 7, 9
SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'


# ---- crc32c (bitwise, reflected 0x82F63B78) + LevelDB mask -----------------------------------------------------------------
def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_T = _crc_table()


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _T[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c_np(arr: np.ndarray) -> int:
    """Same value for large payloads, vectorised per byte position is not possible for a CRC: fall back on 64 KB pieces."""
    return crc32c(arr.tobytes())


def masked(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire format -----------------------------------------------------------------------------------------------------
def varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def f_varint(field: int, v: int) -> bytes:
    return varint(field << 3) + varint(v)


def f_bytes(field: int, payload: bytes) -> bytes:
    return varint((field << 3) | 2) + varint(len(payload)) + payload


def f_fixed32(field: int, v: int) -> bytes:
    return varint((field << 3) | 5) + struct.pack('<I', v)


def entry_proto(dtype, shape, shard, offset, size, crc) -> bytes:
    """BundleEntryProto; proto3 omits zero-valued scalars, as the C++ serializer does."""
    shape_pb = b''.join(f_bytes(2, f_varint(1, int(d))) for d in shape)
    e = f_varint(1, dtype) + f_bytes(2, shape_pb)
    if shard:
        e += f_varint(3, shard)
    if offset:
        e += f_varint(4, offset)
    return e + f_varint(5, size) + f_fixed32(6, masked(crc))


# ---- LevelDB table ----------------------------------------------------------------------------------------------------------------
def shortest_separator(a: bytes, b: bytes) -> bytes:
    """BytewiseComparator::FindShortestSeparator: a <= result < b, as short as the common prefix + one bumped byte allows."""
    n = min(len(a), len(b))
    i = 0
    while i < n and a[i] == b[i]:
        i += 1
    if i < n and a[i] < 0xFF and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def shortest_successor(a: bytes) -> bytes:
    for i, ch in enumerate(a):
        if ch != 0xFF:
            return a[:i] + bytes([ch + 1])
    return a


class Block:
    def __init__(self, interval):
        self.buf, self.restarts, self.n, self.last, self.interval = bytearray(), [0], 0, b'', interval

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.n and self.n % self.interval == 0:
            self.restarts.append(len(self.buf))
        elif self.n:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        self.last, self.n = key, self.n + 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self) -> bytes:
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def write_table(path: str, items, block_size: int):
    items = sorted(items)
    out = bytearray()

    def flush(contents: bytes) -> bytes:
        handle = varint(len(out)) + varint(len(contents))
        out.extend(contents + b'\x00' + struct.pack('<I', masked(crc32c(contents + b'\x00'))))
        return handle

    index = Block(1)
    blk, pending = Block(16), None          # pending = (last key of the flushed block, its handle): the index entry waits for the next key
    for key, value in items:
        if pending is not None:
            index.add(shortest_separator(pending[0], key), pending[1])
            pending = None
        blk.add(key, value)
        if blk.size() >= block_size:
            pending = (key, flush(blk.finish()))
            blk = Block(16)
    if blk.n:
        pending = (blk.last, flush(blk.finish()))
    if pending is not None:
        index.add(shortest_successor(pending[0]), pending[1])
    meta = flush(Block(16).finish())
    idx = flush(index.finish())
    footer = meta + idx
    out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC))
    with open(path, 'wb') as f:
        f.write(out)
    return len(index.restarts)     # = number of data blocks


# ---- the checkpoint -----------------------------------------------------------------------------------------------------------------
def string_scalar(payload: bytes) -> bytes:
    lens = varint(len(payload))
    return lens + struct.pack('<I', masked(crc32c(lens))) + payload


def object_graph(var_paths, full_names, slots):
    """TrackableObjectGraph for variables at attribute paths `var_paths` ("a/b/0/kernel"): one node per path prefix, children
    edges by path component, attributes on the leaves; node "optimizer" carries the slot_variables {original_variable_node_id,
    slot_name, slot_variable_node_id} and has the slot variable nodes as extra objects."""
    ids = {'': 0}
    children = {0: []}
    attrs = {}

    def node(path):
        if path in ids:
            return ids[path]
        parent, _, name = path.rpartition('/')
        pid = node(parent)
        ids[path] = len(ids)
        children[ids[path]] = []
        children[pid].append((ids[path], name))
        return ids[path]

    for p in var_paths:
        attrs[node(p)] = (full_names.get(p, p), p + SUFFIX)
    opt = node('optimizer')
    for n in ('iter', 'beta_1', 'beta_2', 'decay', 'learning_rate'):
        attrs[node('optimizer/' + n)] = ('Adam/' + n, 'optimizer/' + n + SUFFIX)
    attrs[node('save_counter')] = ('save_counter', 'save_counter' + SUFFIX)
    slot_refs = []
    for p, slot in slots:                       # slot variable objects are not reachable by children edges, only by slot_variables
        sid = len(ids)
        ids[f'{p}/.OPTIMIZER_SLOT/optimizer/{slot}'] = sid
        children[sid] = []
        attrs[sid] = (f'Adam/{full_names.get(p, p)}/{slot}', f'{p}/.OPTIMIZER_SLOT/optimizer/{slot}{SUFFIX}')
        slot_refs.append((ids[p], slot, sid))
    nodes = b''
    for nid in range(len(ids)):
        body = b''.join(f_bytes(1, f_varint(1, cid) + f_bytes(2, name.encode())) for cid, name in children[nid])
        if nid in attrs:
            full, key = attrs[nid]
            body += f_bytes(2, f_bytes(1, b'VARIABLE_VALUE') + f_bytes(2, full.encode()) + f_bytes(3, key.encode()))
        if nid == opt:
            body += b''.join(f_bytes(3, f_varint(1, o) + f_bytes(2, s.encode()) + f_varint(3, v)) for o, s, v in slot_refs)
        nodes += f_bytes(1, body)
    return nodes


def write_tf_like_bundle(prefix: str, variables, full_names=None, num_shards: int = 2, block_size: int = 384, seed: int = 0):
    """variables: {attribute path without the VARIABLE_VALUE suffix: float32 array}.  Returns a dict of facts for the test."""
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    rng = np.random.default_rng(seed)
    full_names = full_names or {}
    shards = [bytearray() for _ in range(num_shards)]
    items = [(b'', f_varint(1, num_shards) + f_bytes(3, f_varint(1, 1)))]    # BundleHeaderProto: num_shards, LITTLE endian (0, omitted), version {producer 1}
    turn = 0

    def put(key, dtype, shape, raw):
        nonlocal turn
        sh = turn % num_shards
        turn += 1
        shards[sh].extend(b'\xee' * int(rng.integers(0, 7)))          # padding in front: offsets must be honoured
        items.append((key.encode(), entry_proto(dtype, shape, sh, len(shards[sh]), len(raw), crc32c(raw))))
        shards[sh].extend(raw)

    slots = []
    for path in sorted(variables):
        arr = np.ascontiguousarray(variables[path], dtype='<f4')
        put(path + SUFFIX, DT_FLOAT, arr.shape, arr.tobytes())
        for slot in ('m', 'v'):           # Adam moments: same shape as the variable, different values
            s = (arr * np.float32(0.5 if slot == 'm' else 0.25) + np.float32(1.0)).astype('<f4')
            put(f'{path}/.OPTIMIZER_SLOT/optimizer/{slot}{SUFFIX}', DT_FLOAT, s.shape, s.tobytes())
            slots.append((path, slot))
    for n, v in (('beta_1', 0.9), ('beta_2', 0.999), ('decay', 0.0), ('learning_rate', 1e-4)):
        put(f'optimizer/{n}{SUFFIX}', DT_FLOAT, (), struct.pack('<f', v))
    put(f'optimizer/iter{SUFFIX}', DT_INT64, (), struct.pack('<q', 3000000))
    put(f'save_counter{SUFFIX}', DT_INT64, (), struct.pack('<q', 17))
    graph = object_graph(sorted(variables), full_names, slots)
    put('_CHECKPOINTABLE_OBJECT_GRAPH', DT_STRING, (), string_scalar(graph))
    for i, sh in enumerate(shards):
        with open(f'{prefix}.data-{i:05d}-of-{num_shards:05d}', 'wb') as f:
            f.write(sh)
    nblocks = write_table(prefix + '.index', items, block_size)
    return {'entries': len(items) - 1, 'data_blocks': nblocks, 'shards': num_shards, 'graph_bytes': len(graph)}
