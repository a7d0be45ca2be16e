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
