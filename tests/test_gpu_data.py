"""Device side of the data front end vs the numpy restatement of the reference's _decode_samples
(oracle/m4depth_oracle_data.py).  All arithmetic is + - * / sqrt on float32: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import m4depth_oracle_data as OD
from helpers import F, npy, assert_bits_equal, make_fake_dataset

pytestmark = pytest.mark.gpu


def _decode(dev, raws, kind, out_size, crop=None):
    import m4depth_amd.dataloaders as dl
    name = {0: "midair", 1: "kitti-raw", 2: "tartanair"}[kind]
    loader = dl.get_loader(name)
    loader._set_output_size(out_size=list(out_size))
    loader.device = dev
    loader.usecase = "eval" if crop else "train"
    return loader._decode_device(raws)


@pytest.mark.parametrize("in_size,out_size", [((48, 64), (24, 32)), ((37, 53), (32, 48)), ((32, 32), (32, 32)), ((20, 30), (45, 64))])
def test_rgb_and_midair_depth_decode(dev, in_size, out_size):
    rng = np.random.default_rng(1)
    raws = []
    for _ in range(3):
        depth = rng.uniform(0.5, 90.0, in_size).astype(F)
        raws.append({"rgb": (rng.random(in_size + (3,)) * 255).astype(np.uint8),
                     "depth": (F(512.0) / depth).astype(np.float16).view(np.uint16)})
    rgb, dep = _decode(dev, raws, 0, out_size)
    for i, r in enumerate(raws):
        assert_bits_equal(npy(rgb[i]), OD.decode_rgb(r["rgb"], *out_size), "rgb")
        assert_bits_equal(npy(dep[i]), OD.decode_depth_midair(r["depth"], *out_size), "midair depth")


def test_kitti_and_tartanair_depth_decode(dev):
    rng = np.random.default_rng(2)
    in_size, out_size = (37, 122), (32, 96)
    rgb8 = (rng.random(in_size + (3,)) * 255).astype(np.uint8)
    rgb8[:12, :40] = 0
    sparse = (np.where(rng.random(in_size) > 0.7, rng.uniform(2, 80, in_size), 0.0) * 256).astype(np.uint16)
    for crop in (False, True):
        rgb, dep = _decode(dev, [{"rgb": rgb8, "depth": sparse}], 1, out_size, crop=crop)
        assert_bits_equal(npy(dep[0]), OD.decode_depth_kitti(sparse, *out_size, eval_crop=crop), f"kitti depth crop={crop}")
    dense = rng.uniform(1, 100, in_size).astype(F)
    import m4depth_amd.dataloaders as dl
    loader = dl.get_loader("tartanair")
    loader.in_size = list(in_size)
    loader._set_output_size(out_size=list(out_size))
    loader.device, loader.usecase = dev, "train"
    rgb, dep = loader._decode_device([{"rgb": rgb8, "depth": dense}])
    want_rgb = OD.decode_rgb(rgb8, *out_size)
    assert_bits_equal(npy(rgb[0]), want_rgb, "rgb")
    want = OD.decode_depth_tartanair(dense, want_rgb, *out_size)
    assert_bits_equal(npy(dep[0]), want, "tartanair depth")
    assert (want == 0).any() and (want > 0).any()                          # the black corner is masked


@pytest.mark.parametrize("name", ["midair", "kitti-raw", "tartanair"])
def test_loader_batches_match_the_restatement(dev, tmp_path, name):
    import m4depth_amd.dataloaders as dl
    from PIL import Image
    import os
    size = (48, 64)
    db, rec = make_fake_dataset(str(tmp_path), name, n_traj=2, n_frames=6, size=size)
    loader = dl.get_loader(name)
    if name == "tartanair":
        loader.in_size = list(size)
    out_size = [32, 32] if name == "midair" else [32, 48]
    ds = loader.get_dataset("eval", dl.DataloaderParameters({name: db}, rec, 3, 3, False), batch_size=2, out_size=out_size,
                            device=dev)
    batches = list(ds)
    assert len(batches) == 2
    b0 = batches[0]
    assert tuple(b0["RGB_im"].shape) == (2, 3, out_size[0], out_size[1], 3)
    assert tuple(b0["depth"].shape) == (2, 3, out_size[0], out_size[1], 1)
    assert tuple(b0["rot"].shape) == (2, 3, 4) and tuple(b0["trans"].shape) == (2, 3, 3)
    assert b0["new_traj"].tolist() == [[True, False, False]] * 2
    rows = dl.read_trajectory_csv(os.path.join(rec, "set_0", "traj_0000.csv"))
    for t in range(3):                                                     # batch 0, sequence 0 = trajectory 0, frames 0..2
        with Image.open(os.path.join(db, rows[t]["camera_l"])) as im:
            want_rgb = OD.decode_rgb(np.asarray(im.convert("RGB")), *out_size)
        assert_bits_equal(npy(b0["RGB_im"][0, t]), want_rgb, f"{name} rgb")
        gpath = os.path.join(db, rows[t]["disp" if name == "midair" else "depth"])
        if name == "midair":
            want = OD.decode_depth_midair(np.asarray(Image.open(gpath), np.uint16), *out_size)
        elif name == "kitti-raw":
            want = OD.decode_depth_kitti(np.asarray(Image.open(gpath), np.uint16), *out_size, eval_crop=True)
        else:
            want = OD.decode_depth_tartanair(np.fromfile(gpath, np.float32)[-size[0] * size[1]:].reshape(size), want_rgb, *out_size)
        assert_bits_equal(npy(b0["depth"][0, t]), want, f"{name} depth")
        np.testing.assert_allclose(npy(b0["rot"][0, t]), [rows[t][k] for k in ("qw", "qx", "qy", "qz")], rtol=1e-6)
    if name == "kitti-raw":
        np.testing.assert_allclose(npy(b0["camera"]["f"][0]), [0.58 * out_size[1], 1.92 * out_size[0]], rtol=1e-6)
    else:
        np.testing.assert_allclose(npy(b0["camera"]["c"][0]), [0.5 * out_size[1], 0.5 * out_size[0]], rtol=1e-6)
    # streaming evaluation: single frames, batch 1, no sequence axis
    ds = loader.get_dataset("eval", dl.DataloaderParameters({name: db}, rec, None, 3, False), batch_size=1, out_size=out_size,
                            device=dev)
    frames = list(ds)
    assert len(frames) == 12 and tuple(frames[0]["RGB_im"].shape) == (1, out_size[0], out_size[1], 3)
    assert [bool(f["new_traj"][0]) for f in frames] == ([True] + [False] * 5) * 2
    assert_bits_equal(npy(frames[1]["RGB_im"][0]), npy(b0["RGB_im"][0, 1]), "stream == sequence decode")


def test_flip_augmentation_matches_the_restatement(dev, tmp_path):
    import m4depth_amd.dataloaders as dl
    db, rec = make_fake_dataset(str(tmp_path), "tartanair", n_traj=1, n_frames=3, size=(32, 48))
    loader = dl.get_loader("tartanair")
    loader.in_size = [32, 48]
    loader.get_dataset("eval", dl.DataloaderParameters({"tartanair": db}, rec, 3, 3, False), batch_size=1, out_size=[32, 48],
                       device=dev)
    base = next(iter(loader.dataset))
    sample = {k: npy(base[k][0]) for k in ("RGB_im", "depth", "rot", "trans")}
    sample["camera"] = {k: npy(v[0]) for k, v in base["camera"].items()}

    class Coins:                                                           # the two coin flips of generic.py:232,246
        def __init__(self, vals): self.vals = list(vals)
        def uniform(self, lo, hi): return self.vals.pop(0)

    for v, h in ((True, False), (False, True), (True, True), (False, False)):
        loader.out_data = {"RGB_im": base["RGB_im"][0], "depth": base["depth"][0], "rot": base["rot"][0],
                           "trans": base["trans"][0], "camera": {k: val[0] for k, val in base["camera"].items()}}
        loader.rng = Coins([0.1 if v else 0.9, 0.1 if h else 0.9])
        loader._augmentation_step_flip()
        want = OD.flip_sequence(sample, v, h, 32, 48)
        for k in ("RGB_im", "depth", "rot", "trans"):
            assert_bits_equal(npy(loader.out_data[k]), want[k], f"flip {k} v={v} h={h}")
        assert_bits_equal(npy(loader.out_data["camera"]["c"]), want["camera"]["c"], "flip principal point")


def test_colour_ops(dev):
    from m4depth_amd.dataloaders import color
    rng = np.random.default_rng(3)
    im = torch.from_numpy(rng.random([2, 8, 9, 3]).astype(F)).to(dev)
    np.testing.assert_allclose(npy(color.hsv_to_rgb(color.rgb_to_hsv(im))), npy(im), atol=2e-6)
    np.testing.assert_allclose(npy(color.adjust_hue(im, 0.0)), npy(im), atol=2e-6)
    np.testing.assert_allclose(npy(color.adjust_hue(im, 1.0)), npy(im), atol=1e-5)      # a full turn
    np.testing.assert_allclose(npy(color.adjust_saturation(im, 1.0)), npy(im), atol=2e-6)
    grey = npy(color.adjust_saturation(im, 0.0))
    assert np.allclose(grey[..., 0], grey[..., 1]) and np.allclose(grey[..., 1], grey[..., 2])
    np.testing.assert_allclose(npy(color.adjust_contrast(im, 1.0)), npy(im), atol=1e-6)
    flat = npy(color.adjust_contrast(im, 0.0))
    np.testing.assert_allclose(flat, np.broadcast_to(npy(im).mean(axis=(1, 2), keepdims=True), flat.shape), atol=1e-6)


def test_training_batches_are_augmented_and_consumable(dev, tmp_path):
    """train usecase end to end: augmented Mid-Air batches feed train_step."""
    import m4depth_amd as M
    import m4depth_amd.dataloaders as dl
    from m4depth_amd import synthetic, training as TR
    db, rec = make_fake_dataset(str(tmp_path), "midair", n_traj=2, n_frames=8, size=(64, 64))
    loader = dl.get_loader("midair")
    ds = loader.get_dataset("train", dl.DataloaderParameters({"midair": db}, rec, 4, 3, True), batch_size=2, out_size=[64, 64],
                            device=dev, seed=7)
    model = M.M4Depth(depth_type=loader.depth_type, nbre_levels=2, is_training=True, dscv_range=2, sncv_range=2)
    model.load_numpy_weights(synthetic.init_weights(nbre_levels=2, seed=1, dscv_range=2, sncv_range=2), dev)
    TR.set_trainable(model)
    model.compile(optimizer=torch.optim.Adam(model.parameters(), lr=1e-4, eps=1e-7))
    n = 0
    for batch in ds:
        assert tuple(batch["RGB_im"].shape) == (2, 3, 64, 64, 3) and batch["new_traj"].shape == (2, 3)
        out = model.train_step(batch)
        assert np.isfinite(float(out["loss"]))
        n += 1
    assert n == ds.cardinality() == 2


def test_main_driver_on_a_dataset_in_the_reference_format(dev, tmp_path):
    """python -m m4depth_amd.main --dataset=midair: train (checkpoint), eval on subsequences, streaming eval,
    predict -- the four ways the reference's main.py consumes a dataloader."""
    import json
    import os
    from m4depth_amd import main as MAIN
    db, rec = make_fake_dataset(str(tmp_path), "midair", n_traj=2, n_frames=8, size=(64, 64))
    cfg = os.path.join(str(tmp_path), "datasets_location.json")
    with open(cfg, "w") as fh:
        json.dump({"_comment": "relative to this file", "midair": "./db"}, fh)
    ck = os.path.join(str(tmp_path), "ckpt")
    common = ["--dataset", "midair", "--db_path_config", cfg, "--records_path", rec, "--arch_depth", "2",
              "--height", "64", "--width", "64", "--ckpt_dir", ck]
    assert MAIN.main(["--mode", "train", "--db_seq_len", "4", "--seq_len", "3", "--batch_size", "2", "--epochs", "2"] + common) == 0
    assert os.path.isfile(os.path.join(ck, "train", "ckpt-4.npz"))                 # 2 batches x 2 epochs
    assert MAIN.main(["--mode", "eval", "--db_seq_len", "4", "--batch_size", "2"] + common) == 0
    perfs = np.loadtxt(os.path.join(ck, "perfs-midair.txt"))
    assert perfs.shape == (7,) and np.all(np.isfinite(perfs))
    assert MAIN.main(["--mode", "eval"] + common) == 0                              # streaming, one frame at a time
    stream = np.loadtxt(os.path.join(ck, "perfs-midair.txt"))
    assert np.all(np.isfinite(stream)) and not np.allclose(stream, perfs)
    assert MAIN.main(["--mode", "predict"] + common) == 0
    assert np.load(os.path.join(ck, "predictions.npy")).shape == (16, 64, 64, 1)
