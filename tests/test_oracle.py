"""Oracle self-checks (CPU): the author-derived known-answer test and algebraic
invariants of SURVEY Appendix B, and a pin of the oracle against the committed
golden vectors (which it generated -- parity with the reference itself is
UNPINNED, see oracle/m4depth_oracle.py)."""
import math

import numpy as np
import pytest

from oracle import m4depth_oracle as O
from helpers import F, camera_np, motion_np, assert_bits_equal


# --- KAT-1: one pixel evaluated independently in float64 ---------------------------
def _kat_f64():
    q = np.array([0.9998, 0.01, -0.015, 0.005])
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    fx, fy, cx, cy = 128.0, 64.0, 128.0, 64.0
    t = np.array([0.05, -0.02, 0.4])
    i, j = 200, 20
    mesh = np.array([i + 0.5 - cx, j + 0.5 - cy])
    c2d = np.array([mesh[0] / fx, mesh[1] / fy, 1.0])
    rc = R @ c2d
    alpha = rc[2]
    proj = np.array([rc[0] * fx / alpha, rc[1] * fy / alpha])
    delta = np.array([t[0] * fx - t[2] * proj[0], t[1] * fy - t[2] * proj[1]])
    s = math.hypot(*delta)
    return dict(q=q, R=R, t=t, mesh=mesh, alpha=alpha, proj=proj, delta=delta, s=s, i=i, j=j)


def test_kat1_geometry():
    k = _kat_f64()
    cam = {"f": np.array([[128., 64.]], F), "c": np.array([[128., 64.]], F)}
    q = k["q"].astype(F)[None]
    t = k["t"].astype(F)[None]
    assert np.allclose(O.get_rot_mat(q)[0], k["R"], atol=2e-7)
    m = O.motion_factors(1, 128, 256, q, t, cam)
    j, i = k["j"], k["i"]
    assert abs(m["alpha"][0, j, i] - k["alpha"]) < 1e-6
    assert abs(m["proj_x"][0, j, i] - k["proj"][0]) < 1e-4 and abs(m["proj_y"][0, j, i] - k["proj"][1]) < 1e-4
    assert abs(m["delta_x"][0, j, i] - k["delta"][0]) < 1e-4 and abs(m["sqrt"][0, j, i] - k["s"]) < 1e-4
    # SURVEY Appendix B values
    assert abs(k["alpha"] - 1.0029064968) < 1e-8 and abs(k["s"] - 26.93197566) < 1e-6
    d = np.full([1, 128, 256, 1], 25., F)
    p = O.depth2parallax(d, q, t, cam)
    assert abs(p[0, j, i, 0] - k["s"] / (25 * k["alpha"] + 0.4)) < 1e-6
    assert abs(p[0, j, i, 0] - 1.05728939) < 1e-6
    assert abs(O.parallax2depth(p, q, t, cam)[0, j, i, 0] - 25.0) < 1e-4
    assert abs(O.prev_d2para(d, q, t, cam)[0, j, i, 0] - 1.12845294) < 1e-6


def test_kat1_dscv_queries():
    k = _kat_f64()
    cam = {"f": np.array([[128., 64.]], F), "c": np.array([[128., 64.]], F)}
    q = k["q"].astype(F)[None]
    t = k["t"].astype(F)[None]
    h, w = 128, 256
    rng = np.random.default_rng(0)
    c1 = O.normalize_cuts(rng.standard_normal([1, h, w, 4]).astype(F), 1)
    disp = np.full([1, h, w, 1], 1.05728939, F)
    _, _, y0, x0 = O.get_parallax_sweeping_cv(c1, c1, disp, disp, q, t, cam, 1, 1, return_index=True)
    assert list(y0[0, k["j"], k["i"]]) == [19, 19, 20]
    assert list(x0[0, k["j"], k["i"]]) == [196, 195, 195]
    # float64 re-derivation of the n=0 query: unit steps along the epipolar direction
    u = k["delta"] / k["s"]
    qx = k["i"] + (k["proj"][0] + u[0] * 1.05728939 - k["mesh"][0])
    qy = k["j"] + (k["proj"][1] + u[1] * 1.05728939 - k["mesh"][1])
    assert abs(qx - 195.994112) < 1e-5 and abs(qy - 19.847159) < 1e-5


# --- invariants ------------------------------------------------------------------------
def test_rot_identity_and_orthonormal():
    assert np.array_equal(O.get_rot_mat(np.array([[1, 0, 0, 0]], F))[0], np.eye(3, dtype=F))
    rng = np.random.default_rng(1)
    q = rng.standard_normal([5, 4])
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
    R = O.get_rot_mat(q)
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-6)
    with pytest.raises(ValueError):
        O.get_rot_mat(np.zeros([2, 5], F))


def test_parallax_depth_roundtrip():
    rng = np.random.default_rng(2)
    b, h, w = 2, 24, 40
    rot, trans = motion_np(rng, b)
    cam = camera_np(b, h, w)
    d = (1 + 79 * rng.random([b, h, w, 1])).astype(F)
    back = O.parallax2depth(O.depth2parallax(d, rot, trans, cam), rot, trans, cam)
    assert np.max(np.abs(back - d) / d) < 2e-5


def test_warp_identity_integer_shift_and_border():
    rng = np.random.default_rng(3)
    img = rng.standard_normal([2, 8, 9, 3]).astype(F)
    z = np.zeros([2, 8, 9, 2], F)
    out = O.dense_image_warp(img, z)
    assert_bits_equal(out[:, :-1, :-1], img[:, :-1, :-1], "identity warp interior")   # alpha = 0 exactly
    assert np.allclose(out, img, atol=1e-6)           # last row/col: floor clamps to size-2, alpha = 1
    sh = z.copy()
    sh[..., 0] = 2.0
    sh[..., 1] = -3.0
    out = O.dense_image_warp(img, sh)
    assert_bits_equal(out[:, 0:5, 3:9], img[:, 2:7, 0:6], "integer shift")
    far = z.copy()
    far[..., 0] = -100.0
    far[..., 1] = 100.0
    out = O.dense_image_warp(img, far)
    assert np.allclose(out, img[:, 0:1, 8:9], atol=1e-6)       # border replicate
    # weights form (BackProject) vs lerp form agree to a few ulp
    fl = (rng.standard_normal([2, 8, 9, 2]) * 2).astype(F)
    a = O.dense_image_warp(img, fl)
    bb = O.dense_image_warp(img, fl, use_backproject=True)
    assert np.max(np.abs(a - bb)) < 2e-6


def test_sncv_properties():
    rng = np.random.default_rng(4)
    c = O.normalize_cuts(rng.standard_normal([1, 9, 10, 32]).astype(F), 2)
    sn = O.cost_volume(c, c, 3, nbre_cuts=2)
    assert sn.shape == (1, 9, 10, 98)
    centre = (3 * 7 + 3) * 2
    assert np.allclose(sn[..., centre:centre + 2], 1 / 16, atol=1e-6)     # mean of squares of a unit vector
    assert np.all(sn[0, 0, 0, :3 * 7 * 2] == 0)                           # rows above the image: zero pad
    x = np.array([-1.0, 0.0, 2.0], F)
    assert np.array_equal(O.leaky_relu(x), np.array([-0.1, 0.0, 2.0], F))


def test_dscv_constant_features():
    b, h, w, C, k = 1, 10, 12, 32, 2
    v = np.ones([C], F) / np.sqrt(F(16))
    c = np.tile(v, [b, h, w, 1]).astype(F)
    rng = np.random.default_rng(5)
    rot, trans = motion_np(rng, b)
    disp = (0.5 + rng.random([b, h, w, 1])).astype(F)
    cv, pd = O.get_parallax_sweeping_cv(c, c, disp, disp, rot, trans, camera_np(b, h, w), 4, k)
    assert cv.shape == (b, h, w, 18) and pd.shape == (b, h, w, 9)
    assert np.all(cv == np.float16(1 / 16))
    cv16, _ = O.get_parallax_sweeping_cv(c, c, disp, disp, rot, trans, camera_np(b, h, w), 4, k, cv_accum="fp16_seq")
    assert np.all(cv16 == np.float16(1 / 16))


def test_resize_v1_and_nearest():
    x = np.arange(12, dtype=F).reshape(1, 3, 4, 1)
    up = O.resize_bilinear_v1(x, 6, 8)[0, :, :, 0]
    assert np.array_equal(up[::2, ::2], x[0, :, :, 0])
    assert np.array_equal(up[0, 1::2][:3], (x[0, 0, :3, 0] + x[0, 0, 1:, 0]) / 2)
    assert up[0, 7] == x[0, 0, 3, 0] and up[5, 0] == x[0, 2, 0, 0]
    nn = O.resize_nearest(x, 6, 8)[0, :, :, 0]
    assert np.array_equal(nn, np.repeat(np.repeat(x[0, :, :, 0], 2, 0), 2, 1))


def test_tile_in_batch_order():
    m = np.arange(6).reshape(2, 3)
    t = O.tile_in_batch(m, 3)
    assert t.shape == (6, 3) and np.array_equal(t[2], m[0]) and np.array_equal(t[3], m[1])


def test_conv_same_padding_tf():
    # stride 2 on an even size pads bottom/right only: out[0,0] must not see a virtual row above
    x = np.zeros([1, 4, 4, 1], F)
    x[0, 0, 0, 0] = 1.0
    k = np.zeros([3, 3, 1, 1], F)
    k[0, 0, 0, 0] = 1.0           # top-left tap
    y = O.conv2d_same(x, k, None, 2)
    assert y.shape == (1, 2, 2, 1) and y[0, 0, 0, 0] == 1.0
    y1 = O.conv2d_same(x, k, None, 1)
    assert y1.shape == (1, 4, 4, 1) and y1[0, 1, 1, 0] == 1.0 and y1[0, 0, 0, 0] == 0.0


def test_level_reset_branch_and_f_input_layout():
    from m4depth_amd import synthetic as S
    L = 3
    W = S.init_weights(L, dscv_range=2, sncv_range=2)
    samples, cam = S.make_sequence(1, 2, 64, 96, seed=7)
    model = O.M4Depth(W, L, dscv_range=2, sncv_range=2)
    out, seq = model(samples[:1], cam)
    assert np.all(out["depth"] == 1000.0)                                   # invariant 9
    for l in range(L):
        assert np.all(seq[0][l]["parallax"] == F(2 ** (L - 1 - l)))
        assert np.all(seq[0][l]["other"] == 0)
        assert np.all(model.levels[l].depth_prev_t == 1000.0)
    out, seq = model(samples[1:], cam)
    for l in range(L):
        k = 2 ** ((l + 1) // 2)
        assert model.levels[l].last_f_input.shape[-1] == 5 * k + 1 + 4 + 25 * k + 1 == O.f_input_channels(k, 2, 2)
    assert np.isfinite(out["depth"]).all()


def test_metrics_properties():
    rng = np.random.default_rng(6)
    gt = (1 + 69 * rng.random([2, 16, 16, 1])).astype(F)        # * 1.1 stays below the 80 m clip
    m = O.metrics_batch(gt, gt)
    assert m[0] == 0 and m[1] == 0 and m[2] == 0 and abs(m[3]) < 1e-6 and m[4] == 1
    gt2 = gt.copy()
    gt2[:, :8] = 0.0                                  # masked out
    est = gt * F(1.1)
    assert abs(O.metrics_batch(gt2, est)[0] - 0.1) < 1e-4
    # RMSE_log masks on log(gt) > 1e-6, i.e. drops gt <= ~1
    g = np.array([0.5, 1.0, 2.0, 4.0], F).reshape(1, 2, 2, 1)
    e = g * F(2.0)
    assert abs(O.metrics_batch(g, e)[3] - math.log(2.0)) < 1e-5


# --- golden pin ----------------------------------------------------------------------------
def test_oracle_matches_golden_ops(golden):
    g = golden("ops")
    out, y0, x0 = O.dense_image_warp(g["warp_img"], g["warp_flow"], return_index=True)
    assert_bits_equal(out, g["warp_out"], "warp")
    assert np.array_equal(np.stack([y0, x0], -1), g["warp_idx"])
    assert_bits_equal(O.back_project(g["bp_in"], g["bp_coords"]), g["bp_out"], "back_project")
    cam = camera_np(2, 10, 14)
    for tag in ("q", "e"):
        rot, trans, depth = g[f"cv_{tag}_rot"], g[f"cv_{tag}_trans"], g[f"cv_{tag}_depth"]
        assert_bits_equal(O.depth2parallax(depth, rot, trans, cam), g[f"cv_{tag}_d2p"], "d2p")
        assert_bits_equal(O.prev_d2para(depth, rot, trans, cam), g[f"cv_{tag}_pd2p"], "prev_d2para")
    assert_bits_equal(O.resize_bilinear_v1(g["rs_x"], 9, 13), g["rs_bil_odd"], "resize")
    assert_bits_equal(O.normalize_cuts(g["nm_x"], 4), g["nm_k4"], "normalize")


def test_oracle_matches_golden_cost_volumes(golden):
    g = golden("cost_volumes")
    for tag in "abcd":
        b, h, w, C, k, rd, rs = [int(v) for v in g[f"{tag}_meta"]]
        cam = camera_np(b, h, w)
        cv, pd, y0, x0 = O.get_parallax_sweeping_cv(g[f"{tag}_c1"], g[f"{tag}_c2"], g[f"{tag}_dpt"], g[f"{tag}_disp"],
                                                    g[f"{tag}_rot"], g[f"{tag}_trans"], cam, rd, k, return_index=True)
        assert_bits_equal(cv, g[f"{tag}_cv_fp32_round"], f"dscv {tag}")
        assert np.array_equal(np.stack([y0, x0], -1), g[f"{tag}_idx"])
        assert_bits_equal(O.cost_volume(g[f"{tag}_c1"], g[f"{tag}_c2"], rs, nbre_cuts=k), g[f"{tag}_sncv"], f"sncv {tag}")


def test_oracle_matches_golden_well_conditioned(golden):
    """The well-conditioned model fixture at config-1 size (tests/golden/model_wc_cfg1.npz): the oracle reproduces the
    committed depths (BLAS summation order may differ between hosts: 1e-5, ten times below the tolerance the GPU test
    asserts against these vectors), its float64 evaluation stays within the recorded noise floor, and the fixture is
    what it claims -- s / parallax far from tz on every pixel."""
    from m4depth_amd import synthetic as S
    g = golden("model_wc_cfg1")
    L, rd, rs, H, Wd, T, b, seed = [int(v) for v in g["meta"]]
    W, samples, cam = S.well_conditioned_case(L, b, T, H, Wd, seed, rd, rs)
    _, seq = O.M4Depth(W, L, dscv_range=rd, sncv_range=rs)(samples, cam)
    with O.float64_reference():
        _, seq64 = O.M4Depth(W, L, dscv_range=rd, sncv_range=rs)(samples, cam)
    for l in range(L):
        d, d64 = seq[-1][l]["depth"], seq64[-1][l]["depth"]
        assert np.max(np.abs(d - g[f"l{l}_depth"]) / np.abs(g[f"l{l}_depth"])) < 1e-5
        assert np.max(np.abs(d - d64) / np.abs(d64)) < max(3 * float(g["f32_vs_f64_max_rel_depth"]), 1e-5)
        cam_l = {"f": cam["f"] / F(2.0 ** (l + 1)), "c": cam["c"] / F(2.0 ** (l + 1))}
        m = O.motion_factors(b, H >> (l + 1), Wd >> (l + 1), samples[-1]["rot"], samples[-1]["trans"], cam_l)
        ratio = (m["sqrt"] / seq[-1][l]["parallax"][..., 0]) / np.maximum(np.abs(m["stz"]), 1e-9)
        assert ratio.min() > 20.0, (l, ratio.min())          # depth = (s/para - tz)/alpha never cancels


def test_reference_wiring_crosscheck():
    """the reference's OWN utils/depth_operations.py and utils/dense_image_warp.py, imported unmodified from
    /root/reference under a minimal numpy-backed module named ``tensorflow`` (tests/golden/crosscheck_reference_wiring.py: ~60
    array ops, each one numpy call), return what oracle/m4depth_oracle.py returns on the golden inputs and on seeded inputs --
    warp, get_rot_mat, get_coords_2d, the converters, prev_d2para, reproject (both aux outputs), recompute_depth, tile_in_batch,
    the DSCV (1 / 2 / 4 cuts, ranges 4 and 2) and the SNCV (dilation 1 and 2) -- bit for bit; and the reference's
    m4depth_network.py (FeaturePyramid with DomainNormalization, DispRefiner, DepthEstimatorLevel.call with its temporal memory,
    DepthEstimatorPyramid.call, M4Depth.call) stepped through a reset frame and two full frames of a 3-level model with the
    oracle's weights gives every level's depth / parallax / other of every frame and the final depth of oracle.M4Depth bit for
    bit (the convolution arithmetic is shared: the stand-in's Conv2D calls the oracle's conv2d_same), its metrics.py the seven
    metrics of oracle.metrics_batch to float32 rounding.  It checks the restatement's
    WIRING (reshapes, axes, channel order, operand order) against the reference's text; it does not pin TensorFlow's internal
    arithmetic (the stand-in takes the oracle's documented [UNPINNED] choices), so the parity grade stays "unpinned".  Build
    container only: skipped where /root/reference does not exist (the GPU box)."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "crosscheck_reference_wiring.py")
    spec = importlib.util.spec_from_file_location("crosscheck_reference_wiring", here)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.isfile(os.path.join(mod.REF, "utils", "depth_operations.py")):
        pytest.skip("no /root/reference here (the GPU box): the cross-check is a build-container tool")
    rows = mod.run(verbose=False)
    bad = [(n, m) for n, ok, m in rows if not ok]
    assert len(rows) >= 73 and not bad, bad
    import sys
    assert "tensorflow" not in sys.modules, "the stand-in must not stay installed after the check"
