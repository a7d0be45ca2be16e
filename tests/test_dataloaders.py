"""Host logic of the data front end (no GPU): trajectory csv parsing, the sequence plans that replace
the reference's tf.data graph (dataloaders/generic.py:84-146), and the numpy restatement of the TF
image ops used by _decode_samples."""
import numpy as np
import pytest

from oracle import m4depth_oracle_data as OD
from helpers import make_fake_dataset

F = np.float32


def _loader(name, tmp_path, **kw):
    import m4depth_amd.dataloaders as dl
    db, rec = make_fake_dataset(str(tmp_path), name, **kw)
    loader = dl.get_loader(name)
    return loader, dl, db, rec


def test_csv_reader_types(tmp_path):
    import m4depth_amd.dataloaders as dl
    db, rec = make_fake_dataset(str(tmp_path), "kitti-raw", n_traj=1, n_frames=3)
    rows = dl.read_trajectory_csv(rec + "/set_0/traj_0000.csv")
    assert len(rows) == 3 and rows[0]["id"] == 0 and rows[2]["id"] == 2
    assert isinstance(rows[1]["qw"], float) and isinstance(rows[1]["camera_l"], str) and rows[1]["fx"] == 0.58


def test_train_plan_cuts_shuffles_and_drops_remainders(tmp_path):
    loader, dl, db, rec = _loader("midair", tmp_path, n_traj=3, n_frames=11)
    settings = dl.DataloaderParameters({"midair": db}, rec, 5, 3, True)
    ds = loader.get_dataset("train", settings, batch_size=2, out_size=[32, 32], device="cpu")
    # 3 trajectories x floor(11/5) = 6 chunks -> 3 batches of 2 sequences of 3 frames
    assert ds.cardinality() == 3 and loader.length == 3
    seen = []
    for batch in ds.plan_fn():
        assert len(batch) == 2
        for seq in batch:
            assert len(seq) == 3
            ids = [r["id"] for r in seq]
            assert ids == list(range(ids[0], ids[0] + 3))                  # consecutive frames
            assert ids[0] // 5 == ids[-1] // 5                             # inside one db_seq_len chunk
            seen.append((seq[0]["camera_l"].split("/")[0], ids[0] // 5))
    assert len(set(seen)) == 6                                             # every chunk exactly once per epoch
    # reshuffled each epoch: 6 chunks in 3 batches of 2 -> two epochs coincide with probability 1/720; eight do not
    orders = {tuple((s[0]["camera_l"], s[0]["id"]) for b in ds.plan_fn() for s in b) for _ in range(8)}
    assert len(orders) > 1
    with pytest.raises(Exception):
        loader.get_dataset("train", dl.DataloaderParameters({"midair": db}, rec, 2, 3, True), device="cpu")
    with pytest.raises(Exception):
        loader.get_dataset("train", dl.DataloaderParameters({"midair": db}, rec, None, 3, True), device="cpu")


def test_sharded_plans_give_every_rank_the_same_number_of_disjoint_batches(tmp_path):
    """Data-parallel runs (main.py --mode=train under torchrun): 7 chunks, per-rank batch 1, world 2 -> 3 global batches
    (the odd chunk is dropped on EVERY rank: a rank with one batch more would deadlock the per-step gradient all-reduce),
    the ranks' sequences are disjoint, their union is what one rank with batch 2 would have drawn, and a fixed seed
    reproduces the shuffle."""
    loader, dl, db, rec = _loader("midair", tmp_path, n_traj=1, n_frames=36)
    settings = dl.DataloaderParameters({"midair": db}, rec, 5, 3, True)

    def plan(batch_size, shard, seed=11):
        ld = dl.get_loader("midair")
        ds = ld.get_dataset("train", settings, batch_size=batch_size, out_size=[32, 32], device="cpu", seed=seed, shard=shard)
        return ds.cardinality(), [[(s[0]["camera_l"], s[0]["id"]) for s in b] for b in ds.plan_fn()]

    n0, p0 = plan(1, (0, 2))
    n1, p1 = plan(1, (1, 2))
    ng, pg = plan(2, (0, 1))
    assert n0 == n1 == ng == 3 and len(p0) == len(p1) == 3
    assert all(a + b == g for a, b, g in zip(p0, p1, pg))                  # rank r = slice r of the global batch
    assert not set(sum(p0, [])) & set(sum(p1, []))
    assert plan(1, (0, 2)) == (n0, p0) and plan(1, (0, 2), seed=12)[1] != p0
    # eval on subsequences: contiguous chunks, interleaved by global batch
    ld = dl.get_loader("midair")
    es = dl.DataloaderParameters({"midair": db}, rec, 4, 4, False)
    e0 = list(ld.get_dataset("eval", es, batch_size=2, device="cpu", shard=(0, 2)).plan_fn())
    e1 = list(dl.get_loader("midair").get_dataset("eval", es, batch_size=2, device="cpu", shard=(1, 2)).plan_fn())
    assert len(e0) == len(e1) == 2                                         # 9 chunks // (2 * 2)
    assert [b[0][0]["id"] for b in e0] == [0, 16] and [b[0][0]["id"] for b in e1] == [8, 24]
    with pytest.raises(ValueError):
        dl.get_loader("midair").get_dataset("eval", dl.DataloaderParameters({"midair": db}, rec, None, 4, False),
                                            device="cpu", shard=(0, 2))   # streaming cannot be sharded
    # the plan thread and the augmentation draw from independent generators
    assert ld.plan_rng is not ld.rng


def test_eval_plans(tmp_path):
    loader, dl, db, rec = _loader("kitti-raw", tmp_path, n_traj=2, n_frames=7)
    ds = loader.get_dataset("eval", dl.DataloaderParameters({"kitti-raw": db}, rec, None, 4, False), batch_size=1, device="cpu")
    assert loader.streaming and ds.cardinality() == 14                    # frame stream, batch 1
    plan = list(ds.plan_fn())
    assert [p[0][0]["id"] for p in plan] == list(range(7)) * 2            # trajectories back to back, in order
    ds = loader.get_dataset("eval", dl.DataloaderParameters({"kitti-raw": db}, rec, 3, 2, False), batch_size=2, device="cpu")
    assert not loader.streaming and loader.seq_len == 3                    # db_seq_len overrides seq_len (:121-123)
    assert ds.cardinality() == 2 and all(len(b) == 2 and len(b[0]) == 3 for b in ds.plan_fn())
    assert loader.eval_crop == OD.kitti_eval_crop(256, 768)
    with pytest.raises(Exception):
        loader.get_dataset("nonsense", dl.DataloaderParameters({"kitti-raw": db}, rec, 3, 3, False), device="cpu")


def test_host_decode_returns_the_files_untouched(tmp_path):
    loader, dl, db, rec = _loader("midair", tmp_path, n_traj=1, n_frames=2, size=(24, 40))
    loader.get_dataset("eval", dl.DataloaderParameters({"midair": db}, rec, None, 4, False), device="cpu")
    raw = loader._load_raw(dl.read_trajectory_csv(rec + "/set_0/traj_0000.csv")[1])
    assert raw["rgb"].shape == (24, 40, 3) and raw["rgb"].dtype == np.uint8
    assert raw["depth"].shape == (24, 40) and raw["depth"].dtype == np.uint16
    depth = OD.decode_depth_midair(raw["depth"], 24, 40)[..., 0]
    assert depth.min() > 1.9 and depth.max() < 61.0                        # what make_fake_dataset encoded


def test_oracle_resizes():
    rng = np.random.default_rng(0)
    x = rng.random([6, 8, 3]).astype(F)
    assert np.array_equal(OD.resize_bilinear(x, 6, 8), x)                  # same size: identity
    assert np.array_equal(OD.resize_nearest(x, 6, 8), x)
    want = x.reshape(3, 2, 4, 2, 3).mean(axis=(1, 3))                      # x2 down: 2x2 box mean
    np.testing.assert_allclose(OD.resize_bilinear(x, 3, 4), want, rtol=1e-6)
    assert np.array_equal(OD.resize_nearest(x, 3, 4), x[1::2, 1::2])       # floor((d+.5)*2) = 2d+1
    bits = np.array([[0x4000, 0x3C00]], np.uint16)                         # float16 2.0, 1.0
    assert np.array_equal(OD.decode_depth_midair(bits, 1, 2)[..., 0], np.array([[256.0, 512.0]], F))
    k = OD.decode_depth_kitti(np.array([[512, 0], [256, 1024]], np.uint16), 2, 2, False)[..., 0]
    assert np.array_equal(k, np.array([[2.0, 0.0], [1.0, 4.0]], F))
