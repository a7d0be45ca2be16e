"""Reader for the TensorFlow object-based checkpoints the reference writes and restores
(``tf.train.Checkpoint(model)``, callbacks.py:98-111; the published ``pretrained_weights``), without
TensorFlow: SURVEY section 8, row f-4.

A checkpoint ``<prefix>`` is a *tensor bundle*:
  ``<prefix>.index``                 an immutable sorted string table (the LevelDB table format): key ->
                                     serialized ``BundleEntryProto``; key "" -> ``BundleHeaderProto``
  ``<prefix>.data-SSSSS-of-NNNNN``   the raw little-endian tensor bytes, addressed by (shard, offset, size)
and the key ``_CHECKPOINTABLE_OBJECT_GRAPH`` holds a serialized ``TrackableObjectGraph``: the Python object
tree (children by attribute name, list elements by index) with, per variable, the bundle key of its value.
Variables are therefore looked up by walking ATTRIBUTE PATHS of the reference model
(``encoder/conv_layers_s1/0/kernel`` ...), not by guessing key strings.

Pinning: TensorFlow cannot run here, so the formats are restated from TensorFlow's sources --
core/lib/io/format.cc, block.cc (table), core/util/tensor_bundle (entries, string tensors),
core/protobuf/tensor_bundle.proto and trackable_object_graph.proto.  The table / bundle-entry / tensor-data
layers ARE pinned against files written by TensorFlow itself: the reference ships the bundles of its legacy
model (.legacy/trained_weights/M4Depth-d6), whose index files and the head of one data file are committed
under tests/golden/tf_legacy/; every block trailer and every tensor carries a masked crc32c written by
TensorFlow, and the reader verifies them (tests/test_tf_checkpoint.py).  Those bundles are name-based
(tf.train.Saver), so the object-graph walk (``lookup``) remains UNPINNED: it is tested against a writer that
follows the same description (tests/tf_bundle_writer.py).  Block compression is rejected loudly (the bundle
writer stores index blocks uncompressed).
"""
from __future__ import annotations

import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
VARIABLE_VALUE = "VARIABLE_VALUE"

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DT_STRING = 7


# ------------------------------------------------------------------------------ wire formats
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def parse_proto(buf):
    """Minimal protobuf wire-format walk: {field number: [values]}; varints as int, length-delimited
    fields as bytes, fixed32/64 as int."""
    out = {}
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.setdefault(field, []).append(val)
    return out


def _make_crc_table():
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        table.append(c)
    return table


_CRC_TABLE = _make_crc_table()


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of TensorFlow's core/lib/hash/crc32c.h."""
    crc ^= 0xFFFFFFFF
    table = _CRC_TABLE
    for b in data:
        crc = table[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    """crc32c::Mask: what the table trailers and BundleEntryProto.crc32c store."""
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


class ChecksumError(ValueError):
    pass


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


# ------------------------------------------------------------------------------ the index table
def _read_block(data, offset, size, verify=True):
    if offset + size + 5 > len(data):
        raise ValueError("table block (%d, %d) runs past the end of the file" % (offset, size))
    ctype = data[offset + size]                      # 1-byte compression type, then the masked crc32c of block + type
    if ctype != 0:
        raise NotImplementedError("compressed table block (type %d): tensor-bundle indexes are written uncompressed" % ctype)
    if verify:
        want = struct.unpack_from("<I", data, offset + size + 1)[0]
        if masked_crc32c(data[offset:offset + size + 1]) != want:
            raise ChecksumError("table block at offset %d: crc32c mismatch (corrupt index file)" % offset)
    return data[offset:offset + size]


def _block_entries(block):
    """(key, value) pairs of one table block: prefix-compressed keys, restart array at the end."""
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order; ``verify`` checks every block's
    crc32c trailer."""
    with open(path, "rb") as fh:
        data = fh.read()
    if len(data) < 48:
        raise ValueError(f"{path}: too short for a table file")
    footer = data[-48:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: bad table magic (not a TensorFlow checkpoint index)")
    pos = 0
    _, pos = _varint(footer, pos)                    # metaindex handle (unused)
    _, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size, verify)):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        out.extend(_block_entries(_read_block(data, off, size, verify)))
    return out


# ------------------------------------------------------------------------------ the bundle
class CheckpointReader:
    """``reader = CheckpointReader(prefix)``; ``reader.keys()``, ``reader.tensor(key)``,
    ``reader.lookup("encoder/conv_layers_s1/0/kernel")`` (attribute path from the checkpoint root).
    ``verify`` (default on) checks the crc32c TensorFlow stored for every index block and every tensor read;
    ``data_path`` overrides the ``<prefix>.data-SSSSS-of-NNNNN`` naming (a format string taking shard, shards)."""

    def __init__(self, prefix, verify=True, data_path=None):
        self.prefix = prefix
        self.verify = verify
        self.data_path = data_path
        index = prefix + ".index"
        if not os.path.isfile(index):
            raise FileNotFoundError(f"no checkpoint index at {index}")
        self.entries = {}
        self.num_shards = 1
        for key, value in read_table(index, verify):
            if key == b"":
                hdr = parse_proto(value)             # BundleHeaderProto: num_shards = 1, endianness = 2, version = 3
                self.num_shards = hdr.get(1, [1])[0]
                if hdr.get(2, [0])[0] != 0:
                    raise NotImplementedError("big-endian tensor bundle")
                continue
            e = parse_proto(value)                   # BundleEntryProto
            shape = []
            if 2 in e:
                for dim in parse_proto(e[2][0]).get(2, []):
                    shape.append(_signed(parse_proto(dim).get(1, [0])[0]))
            if 7 in e:
                raise NotImplementedError(f"sliced (partitioned) variable {key!r}")
            self.entries[key.decode()] = dict(dtype=e.get(1, [0])[0], shape=shape, shard=e.get(3, [0])[0],
                                              offset=e.get(4, [0])[0], size=e.get(5, [0])[0],
                                              crc32c=e.get(6, [None])[0])
        self._graph = None

    def keys(self):
        return sorted(self.entries)

    def _bytes(self, e):
        path = (self.data_path or (self.prefix + ".data-%05d-of-%05d")) % (e["shard"], self.num_shards)
        with open(path, "rb") as fh:
            fh.seek(e["offset"])
            raw = fh.read(e["size"])
        if len(raw) != e["size"]:
            raise ValueError(f"{path}: truncated (wanted {e['size']} bytes at {e['offset']})")
        if self.verify and e["crc32c"] is not None and e["dtype"] != DT_STRING:
            if masked_crc32c(raw) != e["crc32c"]:          # string tensors checksum lengths + bytes differently
                raise ChecksumError(f"{path}: crc32c mismatch for the tensor at offset {e['offset']} (corrupt data file)")
        return raw

    def tensor(self, key):
        e = self.entries[key]
        raw = self._bytes(e)
        if e["dtype"] == DT_STRING:                  # [varint64 length] * n, 4-byte crc of the lengths, then the bytes
            n = int(np.prod(e["shape"])) if e["shape"] else 1
            pos, lens = 0, []
            for _ in range(n):
                ln, pos = _varint(raw, pos)
                lens.append(ln)
            pos += 4
            out = []
            for ln in lens:
                out.append(raw[pos:pos + ln])
                pos += ln
            return out[0] if not e["shape"] else np.array(out, dtype=object).reshape(e["shape"])
        if e["dtype"] not in _DTYPES:
            raise NotImplementedError(f"dtype enum {e['dtype']} of {key!r}")
        return np.frombuffer(raw, dtype=_DTYPES[e["dtype"]]).reshape(tuple(e["shape"])).copy()

    # -- object graph --------------------------------------------------------------------------
    def object_graph(self):
        """[{children: {local_name: node_id}, attributes: {name: checkpoint_key}}, ...], node 0 = the root."""
        if self._graph is None:
            if OBJECT_GRAPH_KEY not in self.entries:
                raise KeyError("not an object-based checkpoint (no %s entry)" % OBJECT_GRAPH_KEY)
            nodes = []
            for raw in parse_proto(self.tensor(OBJECT_GRAPH_KEY)).get(1, []):
                node = parse_proto(raw)              # TrackableObject: children = 1, attributes = 2
                children = {}
                for ref in node.get(1, []):
                    r = parse_proto(ref)             # ObjectReference: node_id = 1, local_name = 2
                    children[r.get(2, [b""])[0].decode()] = r.get(1, [0])[0]
                attrs = {}
                for at in node.get(2, []):
                    a = parse_proto(at)              # SerializedTensor: name = 1, full_name = 2, checkpoint_key = 3
                    attrs[a.get(1, [b""])[0].decode()] = a.get(3, [b""])[0].decode()
                nodes.append({"children": children, "attributes": attrs})
            self._graph = nodes
        return self._graph

    def lookup(self, path, attribute=VARIABLE_VALUE):
        """The tensor of the variable reached from the root through ``a/b/0/kernel``-style attribute names."""
        nodes = self.object_graph()
        node = 0
        for name in path.split("/"):
            children = nodes[node]["children"]
            if name not in children:
                raise KeyError(f"{path!r}: no child {name!r} (has: {sorted(children)})")
            node = children[name]
        attrs = nodes[node]["attributes"]
        if attribute not in attrs:
            raise KeyError(f"{path!r} holds no {attribute} (has: {sorted(attrs)})")
        return self.tensor(attrs[attribute])


def latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint: the prefix named by ``model_checkpoint_path`` in <dir>/checkpoint."""
    state = os.path.join(checkpoint_dir, "checkpoint")
    if not os.path.isfile(state):
        return None
    with open(state) as fh:
        m = re.search(r'^model_checkpoint_path:\s*"(.*)"\s*$', fh.read(), re.M)
    if not m:
        return None
    path = m.group(1)
    path = path if os.path.isabs(path) else os.path.join(checkpoint_dir, path)
    return path if os.path.isfile(path + ".index") else None


def model_variable_paths(nbre_levels):
    """numpy-weight name (m4depth_amd.synthetic.init_weights) -> attribute path in the reference model
    (m4depth_network.py:59-74 encoder, :100-114 refiner, :32-36 DINL, :272 levels)."""
    paths = {}
    for i in range(nbre_levels):
        for nm, attr in (("s1", "conv_layers_s1"), ("s2", "conv_layers_s2")):
            paths[f"enc.{nm}.{i}.kernel"] = f"encoder/{attr}/{i}/kernel"
            paths[f"enc.{nm}.{i}.bias"] = f"encoder/{attr}/{i}/bias"
    paths["enc.dn.0.scale"] = "encoder/dn_layers/0/scale"
    paths["enc.dn.0.bias"] = "encoder/dn_layers/0/bias"
    for lvl in range(nbre_levels):
        for j in range(7):
            group, k = ("prep_conv_layers", j) if j < 3 else ("est_d_conv_layers", j - 3)
            for part in ("kernel", "bias"):
                paths[f"lvl.{lvl + 1}.conv.{j}.{part}"] = f"d_estimator/levels/{lvl}/disp_refiner/{group}/{k}/{part}"
    return paths


def load_m4depth_weights(prefix, nbre_levels=6):
    """The weight dict ``M4Depth.load_numpy_weights`` takes, read from a checkpoint of the reference model
    (``tf.train.Checkpoint(model)``: the model is the root object)."""
    reader = CheckpointReader(prefix)
    out = {}
    for name, path in model_variable_paths(nbre_levels).items():
        t = reader.lookup(path)
        out[name] = t.reshape(-1) if name.startswith("enc.dn.") else t       # DINL scale / bias are [1,1,1,C] variables
    return out
