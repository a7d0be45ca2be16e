"""TensorFlow object-based checkpoint reader (m4depth_amd/tf_checkpoint.py) against checkpoints written by
tests/tf_bundle_writer.py in the same format -- no TensorFlow, no real checkpoint here: parity unpinned."""
import os

import numpy as np
import pytest

from m4depth_amd import synthetic, tf_checkpoint as TC
import tf_bundle_writer as W


def _tree(weights, L):
    """The reference model's object tree (m4depth_network.py) filled with ``weights``."""
    enc = {"conv_layers_s1": {}, "conv_layers_s2": {}, "dn_layers": {}}
    for i in range(L):
        for nm, attr in (("s1", "conv_layers_s1"), ("s2", "conv_layers_s2")):
            enc[attr][str(i)] = {"kernel": weights[f"enc.{nm}.{i}.kernel"], "bias": weights[f"enc.{nm}.{i}.bias"]}
    enc["dn_layers"]["0"] = {"scale": weights["enc.dn.0.scale"].reshape(1, 1, 1, -1), "bias": weights["enc.dn.0.bias"].reshape(1, 1, 1, -1)}
    levels = {}
    for lvl in range(L):
        ref = {"prep_conv_layers": {}, "est_d_conv_layers": {}}
        for j in range(7):
            group, k = ("prep_conv_layers", j) if j < 3 else ("est_d_conv_layers", j - 3)
            ref[group][str(k)] = {"kernel": weights[f"lvl.{lvl + 1}.conv.{j}.kernel"], "bias": weights[f"lvl.{lvl + 1}.conv.{j}.bias"]}
        levels[str(lvl)] = {"disp_refiner": ref}
    return {"encoder": enc, "d_estimator": {"levels": levels}, "step_counter": np.array(1234, np.int64)}


@pytest.mark.parametrize("key_style", ["attributes", "opaque"])
def test_round_trip_of_the_model_weights(tmp_path, key_style):
    L = 3
    weights = synthetic.init_weights(nbre_levels=L, seed=11, bias_std=0.1)
    prefix = str(tmp_path / "ckpt-0007")
    W.write_checkpoint(prefix, _tree(weights, L), key_style=key_style)
    got = TC.load_m4depth_weights(prefix, nbre_levels=L)
    assert set(got) == set(weights)
    for k in weights:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], weights[k].reshape(got[k].shape)), k
    reader = TC.CheckpointReader(prefix)
    sc = reader.lookup("step_counter")
    assert sc.shape == () and sc.dtype == np.int64 and sc.item() == 1234                      # an int64 scalar variable
    assert "_CHECKPOINTABLE_OBJECT_GRAPH" in reader.keys() and len(reader.keys()) == len(weights) + 2
    with pytest.raises(KeyError):
        reader.lookup("encoder/conv_layers_s9/0/kernel")
    with pytest.raises(KeyError):
        reader.lookup("encoder")                                          # a container, not a variable


def test_table_blocks_prefix_compression_and_footer(tmp_path):
    items = [(b"", b"hdr")] + [(("key/%03d/suffix" % i).encode(), bytes([i]) * (i % 5 + 1)) for i in range(40)]
    path = str(tmp_path / "t.index")
    W.write_table(path, items, entries_per_block=9)
    assert TC.read_table(path) == sorted(items)
    raw = bytearray(open(path, "rb").read())
    raw[-1] ^= 0xFF
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        TC.read_table(path)                                               # bad magic
    assert W.crc32c(b"123456789") == 0xE3069283                           # the CRC-32C check value


def test_latest_checkpoint_state_file(tmp_path):
    d = str(tmp_path)
    assert TC.latest_checkpoint(d) is None
    weights = synthetic.init_weights(nbre_levels=1, seed=1)
    W.write_checkpoint(os.path.join(d, "ckpt-0003"), _tree(weights, 1))
    with open(os.path.join(d, "checkpoint"), "w") as fh:
        fh.write('model_checkpoint_path: "ckpt-0003"\nall_model_checkpoint_paths: "ckpt-0001"\nall_model_checkpoint_paths: "ckpt-0003"\n')
    assert TC.latest_checkpoint(d) == os.path.join(d, "ckpt-0003")
    with pytest.raises(FileNotFoundError):
        TC.CheckpointReader(os.path.join(d, "ckpt-0009"))


# ------------------------------------------------------------------ files written by TensorFlow itself
# tests/golden/tf_legacy/: the index files of the bundles the reference ships for its legacy model
# (/root/reference/.legacy/trained_weights/M4Depth-d6: M4Depth/features, its optimizers, M4Depth/upscaler, pipeline)
# and the first 66624 bytes (8 tensors) of the features data file.  Every block trailer and every tensor entry holds a
# masked crc32c computed by TensorFlow: they pin the table, BundleEntryProto and tensor-data layers of the reader.
_LEGACY = os.path.join(os.path.dirname(__file__), "golden", "tf_legacy")
_LEGACY_HEAD_BYTES = 66624


@pytest.mark.parametrize("name,n_entries", [("features", 24), ("features_optimizers", 48), ("upscaler", 84), ("pipeline", 22)])
def test_index_files_written_by_tensorflow(name, n_entries):
    kv = TC.read_table(os.path.join(_LEGACY, name + ".index"))           # verifies the crc32c trailer of every block
    assert kv[0][0] == b"" and [k for k, _ in kv] == sorted(k for k, _ in kv)
    assert len(kv) == n_entries + 1
    reader = TC.CheckpointReader(os.path.join(_LEGACY, name))
    assert reader.num_shards == 1 and len(reader.keys()) == n_entries
    for key, e in reader.entries.items():
        itemsize = {1: 4, 9: 8}[e["dtype"]]                               # DT_FLOAT, DT_INT64 (the global step)
        assert e["size"] == itemsize * int(np.prod(e["shape"], dtype=np.int64)), key
        assert e["crc32c"] is not None
    with pytest.raises(KeyError):
        reader.object_graph()                                             # name-based (tf.train.Saver) bundles


def test_index_known_answers():
    reader = TC.CheckpointReader(os.path.join(_LEGACY, "features"))
    e = reader.entries
    assert e["feature_pyramid/layer_1/conv2d_1/kernel"]["shape"] == [3, 3, 3, 16]
    assert e["feature_pyramid/layer_6/conv2d_2/kernel"]["shape"] == [3, 3, 192, 192]
    assert e["feature_pyramid/layer_6/conv2d_2/kernel"]["offset"] == 2761536
    sizes = sorted((v["offset"], v["size"]) for v in e.values())
    assert sizes[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(sizes, sizes[1:]))   # tensors are packed back to back
    assert sizes[-1][0] + sizes[-1][1] == 4088640                         # = the size of the data file in the reference
    up = TC.CheckpointReader(os.path.join(_LEGACY, "upscaler")).entries
    assert up["RIDEN_0/depth_estimator/conv_0/kernel"]["shape"] == [3, 3, 107, 128]
    pipe = TC.CheckpointReader(os.path.join(_LEGACY, "pipeline")).entries
    assert pipe["Pipeline_Global_Step/global_step"]["dtype"] == 9 and pipe["Pipeline_Global_Step/global_step"]["shape"] == []


def _legacy_reader(tmp_path=None, corrupt_at=None):
    data = os.path.join(_LEGACY, "features.data-%05d-of-%05d")
    if corrupt_at is not None:
        raw = bytearray(open(data % (0, 1), "rb").read())
        raw[corrupt_at] ^= 0x01
        data = str(tmp_path / "features.data-%05d-of-%05d")
        open(data % (0, 1), "wb").write(bytes(raw))
    return TC.CheckpointReader(os.path.join(_LEGACY, "features"), data_path=data)


def test_tensor_bytes_match_the_checksums_tensorflow_stored(tmp_path):
    reader = _legacy_reader()
    inside = [k for k, e in reader.entries.items() if e["offset"] + e["size"] <= _LEGACY_HEAD_BYTES]
    assert len(inside) == 8
    for k in inside:
        t = reader.tensor(k)                                              # raises ChecksumError on any wrong byte
        assert t.dtype == np.float32 and list(t.shape) == reader.entries[k]["shape"] and np.isfinite(t).all()
    k1 = reader.tensor("feature_pyramid/layer_1/conv2d_1/kernel")
    assert 0.1 < float(np.abs(k1).max()) < 2.0 and float(np.abs(k1).mean()) > 1e-3     # trained weights, not padding
    outside = "feature_pyramid/layer_3/conv2d_1/bias"
    with pytest.raises(ValueError):
        reader.tensor(outside)                                            # beyond the committed head: truncated, loudly
    bad = _legacy_reader(tmp_path, corrupt_at=64 + 100)                   # one bit inside layer_1/conv2d_1/kernel
    with pytest.raises(TC.ChecksumError):
        bad.tensor("feature_pyramid/layer_1/conv2d_1/kernel")
    assert np.array_equal(bad.tensor("feature_pyramid/layer_1/conv2d_1/bias"), reader.tensor("feature_pyramid/layer_1/conv2d_1/bias"))
    unchecked = TC.CheckpointReader(os.path.join(_LEGACY, "features"), verify=False, data_path=bad.data_path)
    assert unchecked.tensor("feature_pyramid/layer_1/conv2d_1/kernel").shape == (3, 3, 3, 16)


def test_corrupt_index_block_is_rejected(tmp_path):
    raw = bytearray(open(os.path.join(_LEGACY, "features.index"), "rb").read())
    raw[40] ^= 0x10
    path = str(tmp_path / "bad.index")
    open(path, "wb").write(bytes(raw))
    with pytest.raises(TC.ChecksumError):
        TC.read_table(path)


def test_checksum_functions():
    assert TC.crc32c(b"123456789") == 0xE3069283
    assert TC.crc32c(b"\x00" * 32) == 0x8A9136AA and TC.crc32c(b"\xff" * 32) == 0x62A8AB43       # RFC 3720 B.4
    assert TC.masked_crc32c(b"abc") == W.masked_crc(b"abc")


_REFERENCE_LEGACY = "/root/reference/.legacy/trained_weights/M4Depth-d6"


@pytest.mark.skipif(not os.path.isdir(_REFERENCE_LEGACY), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("prefix", ["M4Depth/features/checkpoint-200000", "pipeline/checkpoint-200000"])
def test_complete_bundles_of_the_reference(prefix):
    """Every tensor of the two bundles whose data files the reference ships (8 MB), crc-verified."""
    assert TC.latest_checkpoint(os.path.join(_REFERENCE_LEGACY, os.path.dirname(prefix))) == os.path.join(_REFERENCE_LEGACY, prefix)
    reader = TC.CheckpointReader(os.path.join(_REFERENCE_LEGACY, prefix))
    total = 0
    for k in reader.keys():
        t = reader.tensor(k)
        assert list(t.shape) == reader.entries[k]["shape"]
        total += t.nbytes
    assert total == os.path.getsize(os.path.join(_REFERENCE_LEGACY, prefix) + ".data-00000-of-00001")
