"""Test-side WRITER of TensorFlow tensor-bundle checkpoints (the format m4depth_amd/tf_checkpoint.py reads),
following the same description of TensorFlow's on-disk formats: LevelDB-style table with prefix-compressed
blocks, restart arrays, masked crc32c trailers and the 48-byte footer; BundleHeaderProto / BundleEntryProto;
string-tensor encoding; TrackableObjectGraph.  It exists because no TensorFlow and no real checkpoint are
available in the build environment: reader and writer pin each other, nothing more (parity unpinned)."""
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def field(num, wt, payload):
    if wt == 0:
        return varint(num << 3) + varint(payload)
    if wt == 2:
        return varint((num << 3) | 2) + varint(len(payload)) + payload
    if wt == 5:
        return varint((num << 3) | 5) + struct.pack("<I", payload)
    raise ValueError(wt)


def build_block(items, restart_interval=16):
    out = bytearray()
    restarts = []
    prev = b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, items, entries_per_block=7):
    items = sorted(items)
    data = bytearray()
    index_items = []

    def emit(block):
        off = len(data)
        data.extend(block)
        data.extend(b"\x00" + struct.pack("<I", masked_crc(block + b"\x00")))   # type 0 = uncompressed, then the crc
        return varint(off) + varint(len(block))

    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        handle = emit(build_block(chunk))
        index_items.append((chunk[-1][0] + b"\x00", handle))                     # a key >= every key of the block
    meta = emit(build_block([]))
    index = emit(build_block(index_items, restart_interval=1))
    footer = meta + index
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    data.extend(footer)
    with open(path, "wb") as fh:
        fh.write(bytes(data))


def entry_proto(dtype_enum, shape, offset, size, crc):
    dims = b"".join(field(2, 2, field(1, 0, d)) for d in shape)
    return (field(1, 0, dtype_enum) + field(2, 2, dims) + field(3, 0, 0) + field(4, 0, offset) + field(5, 0, size)
            + field(6, 5, crc))


def object_graph_proto(tree, key_of):
    """tree: nested dicts; a leaf is the name of a variable -> TrackableObjectGraph bytes.  ``key_of(path)`` names
    the bundle key of the variable at ``path``."""
    nodes = []

    def add(node, path):
        idx = len(nodes)
        nodes.append(None)
        children, attrs = [], b""
        if isinstance(node, dict):
            for name, child in node.items():
                cid = add(child, path + [name])
                children.append(field(1, 2, field(1, 0, cid) + field(2, 2, name.encode())))
        else:                                            # a variable: one serialized tensor
            attrs = field(2, 2, field(1, 2, b"VARIABLE_VALUE") + field(2, 2, "/".join(path).encode())
                          + field(3, 2, key_of(path).encode()))
        nodes[idx] = b"".join(children) + attrs
        return idx

    add(tree, [])
    return b"".join(field(1, 2, n) for n in nodes)


def write_checkpoint(prefix, tree_values, key_style="attributes", extra_root_children=None):
    """tree_values: nested dicts whose leaves are numpy arrays = the object tree of the saved model."""
    def key_of(path):
        if key_style == "attributes":
            return "/".join(path) + "/.ATTRIBUTES/VARIABLE_VALUE"
        return "var_%08x/.ATTRIBUTES/VARIABLE_VALUE" % crc32c("/".join(path).encode())     # unrelated to the attribute path

    flat = {}

    def walk(node, path):
        if isinstance(node, dict):
            for k, v in node.items():
                walk(v, path + [k])
        else:
            flat[key_of(path)] = np.asarray(node)
    walk(tree_values, [])
    shape_tree = tree_values if not extra_root_children else dict(tree_values, **extra_root_children)

    def names(node):
        return {k: names(v) for k, v in node.items()} if isinstance(node, dict) else "var"
    graph = object_graph_proto(names(shape_tree), key_of)
    for path_children in (extra_root_children or {}).values():
        walk(path_children, ["__extra__"])

    blob = bytearray()
    items = [(b"", field(1, 0, 1) + field(2, 0, 0) + field(3, 2, field(1, 0, 1)))]      # header: 1 shard, little endian
    for key in sorted(flat):
        arr = flat[key]
        raw = arr.tobytes()
        items.append((key.encode(), entry_proto(_DT[arr.dtype], arr.shape, len(blob), len(raw), masked_crc(raw))))
        blob += raw
    lens = varint(len(graph))
    sraw = lens + struct.pack("<I", masked_crc(lens)) + graph                           # scalar DT_STRING tensor
    items.append((b"_CHECKPOINTABLE_OBJECT_GRAPH", entry_proto(7, (), len(blob), len(sraw), masked_crc(sraw))))
    blob += sraw
    with open(prefix + ".data-00000-of-00001", "wb") as fh:
        fh.write(bytes(blob))
    write_table(prefix + ".index", items)
