"""HIP kernels (through the C ABI / the reference-named Python ops) vs the CPU
oracle and the committed golden vectors.

Tolerances (written here, as the contract demands):
  * int32 bilinear index grids ........................ bit-exact
  * warps, back_project, converters, DSCV cv, SNCV,
    normalisation, resizes (+,-,*,/,sqrt only) ........ bit-exact (float32 bit pattern)
  * anything through exp/log (libm differs) ........... 2e-6 relative
  * back_project_grad scatter (atomic order) .......... 1e-5 absolute
"""
import numpy as np
import pytest
import torch

from oracle import m4depth_oracle as O
from helpers import F, camera_np, motion_np, to_dev, npy, assert_bits_equal, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import m4depth_amd
    return m4depth_amd


# ------------------------------------------------------------------------------- golden
def test_golden_warp_and_backproject(M, dev, golden):
    g = golden("ops")
    out, idx = M.dense_image_warp(to_dev(g["warp_img"], dev), to_dev(g["warp_flow"], dev), return_index=True)
    assert np.array_equal(npy(idx), g["warp_idx"])                    # bit-exact index grid
    assert_bits_equal(npy(out), g["warp_out"], "dense_image_warp")
    bp = M.back_project(to_dev(g["bp_in"], dev), to_dev(g["bp_coords"], dev))
    assert_bits_equal(npy(bp), g["bp_out"], "back_project")
    gi, gc = M.back_project_grad(to_dev(g["bp_in"], dev), to_dev(g["bp_coords"], dev), to_dev(g["bp_grad"], dev))
    assert_bits_equal(npy(gc), g["bp_gco"], "back_project_grad coords")
    assert np.max(np.abs(npy(gi) - g["bp_gin"])) < 1e-5


def test_golden_converters_and_reproject(M, dev, golden):
    g = golden("ops")
    cam = to_dev(camera_np(2, 10, 14), dev)
    for tag in ("q", "e"):
        rot, trans, depth = [to_dev(g[f"cv_{tag}_{k}"], dev) for k in ("rot", "trans", "depth")]
        para = M.depth2parallax(depth, rot, trans, cam)
        assert_bits_equal(npy(para), g[f"cv_{tag}_d2p"], "depth2parallax")
        assert_bits_equal(npy(M.parallax2depth(para, rot, trans, cam)), g[f"cv_{tag}_p2d"], "parallax2depth")
        assert_bits_equal(npy(M.prev_d2para(depth, rot, trans, cam)), g[f"cv_{tag}_pd2p"], "prev_d2para")
        assert_bits_equal(npy(M.recompute_depth(depth, rot, trans, cam)), g[f"cv_{tag}_recompute"], "recompute_depth")
    out, (pmr, rotc) = M.reproject(to_dev(g["rp_map"], dev), to_dev(g["rp_depth"], dev), to_dev(g["rp_rot"], dev),
                                   to_dev(g["rp_trans"], dev), cam)
    assert_bits_equal(npy(out), g["rp_out"], "reproject")
    assert_bits_equal(npy(pmr), g["rp_pmr"], "reproject aux 0")
    assert_bits_equal(npy(rotc), g["rp_rotc"], "reproject aux 1")


def test_golden_resize_normalize(M, dev, golden):
    from m4depth_amd import network_ops as nops
    g = golden("ops")
    x = to_dev(g["rs_x"], dev)
    assert_bits_equal(npy(nops.resize_bilinear_v1(x, 10, 14)), g["rs_bil_x2"], "resize x2")
    assert_bits_equal(npy(nops.resize_bilinear_v1(x, 9, 13)), g["rs_bil_odd"], "resize odd")
    assert_bits_equal(npy(nops.resize_nearest(x, 10, 14)), g["rs_near_x2"], "nearest x2")
    assert_bits_equal(npy(nops.resize_nearest(x, 11, 15)), g["rs_near_odd"], "nearest odd")
    nx = to_dev(g["nm_x"], dev)
    for k in (1, 4, 3):
        assert_bits_equal(npy(nops.normalize_cuts(nx, k)), g[f"nm_k{k}"], f"normalize k={k}")


@pytest.mark.parametrize("tag", list("abcd"))
def test_golden_cost_volumes(M, dev, golden, tag):
    g = golden("cost_volumes")
    b, h, w, C, k, rd, rs = [int(v) for v in g[f"{tag}_meta"]]
    cam = to_dev(camera_np(b, h, w), dev)
    c1, c2, dpt, disp, rot, trans = [to_dev(g[f"{tag}_{n}"], dev) for n in ("c1", "c2", "dpt", "disp", "rot", "trans")]
    for acc in ("fp32_round", "fp16_seq"):
        cv, pd, idx = M.get_parallax_sweeping_cv(c1, c2, dpt, disp, rot, trans, cam, rd, k, cv_accum=acc,
                                                 return_index=True)
        assert np.array_equal(npy(idx), g[f"{tag}_idx"]), "DSCV index grid must be bit-exact"
        assert_bits_equal(npy(cv), g[f"{tag}_cv_{acc}"], f"dscv cv {acc}")
        assert_bits_equal(npy(pd), g[f"{tag}_prev_disp"], "dscv prev_disp")
    assert_bits_equal(npy(M.cost_volume(c1, c2, rs, nbre_cuts=k)), g[f"{tag}_sncv"], "sncv")
    assert_bits_equal(npy(M.cost_volume(c1, c1, rs, nbre_cuts=k)), g[f"{tag}_sncv_auto"], "sncv auto")
    if tag == "a":
        assert_bits_equal(npy(M.cost_volume(c1, c2, 2, dilation_rate=2, nbre_cuts=k)), g["a_sncv_dil2"], "sncv dil 2")


# ---------------------------------------------------------------- seeded, against the oracle
LEVEL_GEOM = [(1, 48, 80, 16, 1), (2, 24, 40, 32, 2), (3, 24, 40, 64, 2), (4, 12, 20, 96, 4), (5, 12, 20, 128, 4),
              (6, 6, 20, 192, 8)]


@pytest.mark.parametrize("lvl,h,w,C,k", LEVEL_GEOM)
def test_dscv_sncv_level_geometries(M, dev, lvl, h, w, C, k):
    """Every (C, cuts) pair of the 6-level pyramid, batch 2, DSCV r=4 / SNCV r=3."""
    rng = np.random.default_rng(100 + lvl)
    b = 2
    cam = camera_np(b, h, w)
    rot, trans = motion_np(rng, b, t_scale=(3.0, 3.0, 1.0))
    c1 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), k)
    c2 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), k)
    disp = (0.2 + 6.0 * rng.random([b, h, w, 1])).astype(F)
    dpt = (0.2 + 6.0 * rng.random([b, h, w, 1])).astype(F)
    ocv, opd, oy, ox = O.get_parallax_sweeping_cv(c1, c2, dpt, disp, rot, trans, cam, 4, k, return_index=True)
    cv, pd, idx = M.get_parallax_sweeping_cv(*to_dev([c1, c2, dpt, disp, rot, trans], dev), to_dev(cam, dev), 4, k,
                                             return_index=True)
    assert np.array_equal(npy(idx), np.stack([oy, ox], -1))
    assert_bits_equal(npy(cv), ocv, "dscv")
    assert_bits_equal(npy(pd), opd, "prev_disp")
    sn = M.cost_volume(to_dev(c1, dev), to_dev(c1, dev), 3, nbre_cuts=k)
    assert_bits_equal(npy(sn), O.cost_volume(c1, c1, 3, nbre_cuts=k), "sncv")


def test_warp_edge_cases(M, dev):
    rng = np.random.default_rng(7)
    img = rng.standard_normal([1, 2, 2, 1]).astype(F)                 # minimum legal size
    fl = (rng.standard_normal([1, 2, 2, 2]) * 2).astype(F)
    assert_bits_equal(npy(M.dense_image_warp(to_dev(img, dev), to_dev(fl, dev))), O.dense_image_warp(img, fl), "2x2")
    img3 = rng.standard_normal([5, 6, 3]).astype(F)                   # rank-3 image (dense_image_warp.py:229-232)
    fl3 = (rng.standard_normal([1, 5, 6, 2])).astype(F)
    got = M.dense_image_warp(to_dev(img3, dev), to_dev(fl3, dev))
    assert got.shape == (5, 6, 3)
    assert_bits_equal(npy(got), O.dense_image_warp(img3[None], fl3)[0], "rank 3")
    with pytest.raises(ValueError):
        M.dense_image_warp(to_dev(np.zeros([1, 1, 4, 2], F), dev), to_dev(np.zeros([1, 1, 4, 2], F), dev))
    # the BackProject route of dense_image_warp.py:246-253
    import m4depth_amd.dense_image_warp  # noqa: F401
    import sys
    mod = sys.modules["m4depth_amd.dense_image_warp"]
    img = rng.standard_normal([2, 7, 9, 4]).astype(F)
    fl = (rng.standard_normal([2, 7, 9, 2]) * 4).astype(F)
    mod.use_cuda_backproject = True
    try:
        got = mod.dense_image_warp(to_dev(img, dev), to_dev(fl, dev))
    finally:
        mod.use_cuda_backproject = False
    assert_bits_equal(npy(got), O.dense_image_warp(img, fl, use_backproject=True), "op route")


def test_interpolate_bilinear_free_points(M, dev):
    rng = np.random.default_rng(8)
    grid = rng.standard_normal([2, 6, 7, 3]).astype(F)
    q = (rng.random([2, 11, 2]) * np.array([7.0, 8.0]) - 1.0).astype(F)
    out, idx = M._interpolate_bilinear(to_dev(grid, dev), to_dev(q, dev), return_index=True)
    o, y0, x0 = O.interpolate_bilinear(grid, q, return_index=True)
    assert np.array_equal(npy(idx), np.stack([y0, x0], -1))
    assert_bits_equal(npy(out), o, "interpolate_bilinear ij")
    out_xy = M._interpolate_bilinear(to_dev(grid, dev), to_dev(q[..., ::-1].copy(), dev), indexing='xy')
    assert_bits_equal(npy(out_xy), o, "interpolate_bilinear xy")
    with pytest.raises(ValueError):
        M._interpolate_bilinear(to_dev(grid, dev), to_dev(q, dev), indexing='zz')


def test_backproject_autograd(M, dev):
    rng = np.random.default_rng(9)
    inp = torch.tensor(rng.standard_normal([1, 5, 6, 1, 3]).astype(F), device=dev, requires_grad=True)
    co = torch.tensor((rng.random([1, 5, 6, 2, 1, 2]) * np.array([5.0, 4.0])).astype(F), device=dev, requires_grad=True)
    out = M.back_project(inp, co)
    g = torch.tensor(rng.standard_normal(list(out.shape)).astype(F), device=dev)
    out.backward(g)
    gi, gc = O.back_project_grad(npy(inp), npy(co), npy(g))
    assert np.max(np.abs(npy(inp.grad) - gi)) < 1e-5
    assert_bits_equal(npy(co.grad), gc, "coords grad")


def test_error_behaviour(M, dev):
    z = torch.zeros([1, 4, 4, 1], device=dev)
    cam = to_dev(camera_np(1, 4, 4), dev)
    with pytest.raises(ValueError):                                    # get_rot_mat, depth_operations.py:53
        M.parallax2depth(z, torch.zeros([1, 5], device=dev), torch.zeros([1, 3], device=dev), cam)
    with pytest.raises(ValueError):                                    # reproject, depth_operations.py:78-79
        M.reproject(torch.zeros([1, 4, 4, 2], device=dev), torch.zeros([1, 3, 4, 1], device=dev),
                    torch.zeros([1, 4], device=dev), torch.zeros([1, 3], device=dev), cam)
    with pytest.raises(RuntimeError):                                  # no CPU fallback
        M.cost_volume(torch.zeros([1, 4, 4, 4]), torch.zeros([1, 4, 4, 4]), 1)
    with pytest.raises(ValueError):
        M.cost_volume(torch.zeros([1, 4, 4, 6], device=dev), torch.zeros([1, 4, 4, 6], device=dev), 1, nbre_cuts=4)


def test_level_pre_post_kernels(M, dev):
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(10)
    b, h, w = 2, 12, 20
    rot, trans = motion_np(rng, b)
    cam = camera_np(b, h, w)
    prev = {"depth": (1 + 50 * rng.random([b, 6, 10, 1])).astype(F),
            "parallax": (0.1 + 3 * rng.random([b, 6, 10, 1])).astype(F),
            "other": rng.standard_normal([b, 6, 10, 4]).astype(F)}
    dprev = (1 + 50 * rng.random([b, h, w, 1])).astype(F)
    f_in = torch.full((b, h, w, 12), -7.0, device=dev)
    para, depth, other, para_t = nops.level_pre(to_dev(prev, dev), to_dev(dprev, dev), to_dev(trans, dev),
                                                to_dev(cam, dev), b, h, w, dev, f_input=f_in, log_off=3,
                                                other_off=4, log_scale=0.25)
    op = O.resize_bilinear_v1(prev["parallax"], h, w) * F(2.)
    assert_bits_equal(npy(para), op, "para_prev_l")
    assert_bits_equal(npy(depth), O.resize_bilinear_v1(prev["depth"], h, w), "depth_prev_l")
    assert_bits_equal(npy(other), O.resize_bilinear_v1(prev["other"], h, w), "other_prev_l")
    assert_bits_equal(npy(para_t), O.prev_d2para(dprev, rot, trans, cam), "para_prev_t")
    fi = npy(f_in)
    assert np.max(rel_err(fi[..., 3], np.log(op[..., 0] * F(0.25)), 1e-3)) < 2e-6
    assert_bits_equal(fi[..., 4:8], npy(other), "f_input other")
    assert np.all(fi[..., :3] == -7.0) and np.all(fi[..., 8:] == -7.0)          # untouched channels
    # coarsest level: constants (m4depth_network.py:198-200)
    para, depth, other, _ = nops.level_pre(None, None, None, None, b, h, w, dev)
    assert torch.all(para == 1) and torch.all(depth == 1000) and torch.all(other == 0)
    # tail
    ro = (rng.standard_normal([b, h, w, 5]) * 4).astype(F)
    state = torch.empty((b, h, w, 1), device=dev)
    p, d, o = nops.level_post(to_dev(ro, dev), to_dev(rot, dev), to_dev(trans, dev), to_dev(cam, dev), 4.0,
                              depth_state=state)
    ep = (np.exp(np.clip(ro[..., :1], F(-7), F(7))) / F(4.0)).astype(F)
    assert np.max(rel_err(npy(p), ep)) < 2e-6
    assert_bits_equal(npy(d), O.parallax2depth(npy(p), rot, trans, cam), "depth from the kernel's own parallax")
    assert_bits_equal(npy(state), npy(d), "depth state")
    assert_bits_equal(npy(o), ro[..., 1:], "other")


# --------------------------------------- full-size properties (BASELINE sizes, no oracle run)
def test_fullsize_properties(M, dev):
    """384x1280-pyramid level-1 geometry (192x640, C=16), batch 4: size-independent
    properties instead of an oracle run."""
    torch.manual_seed(0)
    b, h, w, C = 4, 192, 640, 16
    rng = np.random.default_rng(11)
    rot, trans = motion_np(rng, b)
    cam = to_dev(camera_np(b, h, w), dev)
    rot, trans = to_dev(rot, dev), to_dev(trans, dev)
    from m4depth_amd import network_ops as nops
    f = nops.normalize_cuts(torch.randn(b, h, w, C, device=dev), 1)
    assert torch.allclose((f * f).sum(-1), torch.ones(b, h, w, device=dev), atol=1e-5)
    # SNCV: centre displacement of a unit vector = 1/C; out-of-image displacements = 0
    sn = M.cost_volume(f, f, 3, nbre_cuts=1)
    assert torch.allclose(sn[..., 24], torch.full((b, h, w), 1.0 / C, device=dev), atol=1e-6)
    assert torch.all(sn[:, 0, :, :21] == 0) and torch.all(sn[:, :, 0, 0::7] == 0)
    # symmetry of the autocorrelation: cost(p, p+d) == cost(p+d, p) before the leaky relu sign is shared
    assert torch.equal(sn[:, 5:-5, 5:-5, 24 + 1], sn[:, 5:-5, 6:-4, 24 - 1])
    # identity warp / integer shift (bitwise, interior)
    z = torch.zeros(b, h, w, 2, device=dev)
    out, idx = M.dense_image_warp(f, z, return_index=True)
    assert torch.equal(out[:, :-1, :-1], f[:, :-1, :-1])
    jj, ii = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    assert torch.equal(idx[0, :-1, :-1, 0], jj[:-1, :-1].int()) and torch.equal(idx[0, :-1, :-1, 1], ii[:-1, :-1].int())
    # parallax <-> depth round trip
    d = 1 + 79 * torch.rand(b, h, w, 1, device=dev)
    back = M.parallax2depth(M.depth2parallax(d, rot, trans, cam), rot, trans, cam)
    assert torch.max(torch.abs(back - d) / d) < 5e-5
    # DSCV of spatially constant unit features = fp16(1/C) for every hypothesis
    const = torch.full((b, h, w, C), 0.25, device=dev)
    disp = 0.5 + 3 * torch.rand(b, h, w, 1, device=dev)
    cv, pd = M.get_parallax_sweeping_cv(const, const, disp, disp, rot, trans, cam, 4, 1)
    assert torch.all(cv == float(np.float16(1.0 / C)))
    # hypotheses step by exactly one pixel along the epipolar line: the warped parallax map of a
    # constant map is that constant
    cst = torch.full((b, h, w, 1), 2.5, device=dev)
    _, pd = M.get_parallax_sweeping_cv(f, f, cst, disp, rot, trans, cam, 4, 1)
    assert torch.allclose(pd, torch.full_like(pd, 2.5), atol=1e-6)


def test_bias_act_epilogue(M, dev):
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(12)
    for C in (16, 5, 128):
        x = rng.standard_normal([2, 7, 9, C]).astype(F)
        b = rng.standard_normal([C]).astype(F)
        for slope in (0.1, 1.0):
            got = nops.bias_act_(to_dev(x.copy(), dev), to_dev(b, dev), slope)
            y = x + b
            ref = np.where(y > 0, y, y * F(slope)).astype(F)
            assert_bits_equal(npy(got), ref, f"bias_act C={C} slope={slope}")


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_dscv_kernel_variants_bit_identical(M, dev, variant):
    """generic / wave / LDS-window / hypothesis-per-lane DSCV kernels: same bits, incl. at a
    size whose tiles straddle the image border and with large flows (window fallback)."""
    from m4depth_amd._lib import lib
    rng = np.random.default_rng(300)
    try:
        lib.m4d_dscv_set_variant(variant)
        for (b, h, w, C, k, tscale) in [(2, 40, 72, 16, 1, 3.0), (1, 24, 40, 32, 2, 3.0), (1, 16, 24, 64, 2, 3.0),
                                        (1, 20, 44, 16, 1, 40.0)]:
            cam = camera_np(b, h, w)
            rot, trans = motion_np(rng, b, t_scale=(tscale, tscale, 1.0))
            c1 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), k)
            c2 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), k)
            disp = (0.2 + 6.0 * rng.random([b, h, w, 1])).astype(F)
            dpt = (0.2 + 6.0 * rng.random([b, h, w, 1])).astype(F)
            ocv, opd, oy, ox = O.get_parallax_sweeping_cv(c1, c2, dpt, disp, rot, trans, cam, 4, k, return_index=True)
            cv, pd, idx = M.get_parallax_sweeping_cv(*to_dev([c1, c2, dpt, disp, rot, trans], dev), to_dev(cam, dev), 4, k,
                                                     return_index=True)
            assert np.array_equal(npy(idx), np.stack([oy, ox], -1))
            assert_bits_equal(npy(cv), ocv, f"dscv variant {variant} C={C}")
            assert_bits_equal(npy(pd), opd, f"prev_disp variant {variant}")
    finally:
        lib.m4d_dscv_set_variant(1)


def test_fused_dinl_and_metrics(M, dev):
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(13)
    x = (rng.standard_normal([2, 40, 56, 16]) * 2 + 0.5).astype(F)
    scale = (1 + 0.1 * rng.standard_normal(16)).astype(F)
    bias = (0.1 * rng.standard_normal(16)).astype(F)
    ref = O.domain_normalization(x, scale, bias)
    got = nops.dinl_act(to_dev(x, dev), to_dev(scale, dev), to_dev(bias, dev), 1.0)
    assert np.max(np.abs(npy(got) - ref)) < 2e-6                       # reductions in a different (fixed) order
    got = nops.dinl_act(to_dev(x, dev), to_dev(scale, dev), to_dev(bias, dev), 0.1)
    assert np.max(np.abs(npy(got) - O.leaky_relu(ref, 0.1))) < 2e-6
    a = nops.dinl_act(to_dev(x, dev), to_dev(scale, dev), to_dev(bias, dev), 0.1)
    assert torch.equal(a, got)                                          # deterministic
    gt = (80 * rng.random([2, 64, 96, 1])).astype(F)
    gt[0, :7] = 0.0
    gt[1, 5, :9] = 200.0                                                # above the 80 m clip
    est = (gt * (1 + 0.2 * rng.standard_normal(gt.shape)) + 0.01).astype(F)
    est[0, 20, :5] = -3.0                                               # below the 0.001 clip
    vals = npy(nops.depth_metrics(to_dev(gt, dev), to_dev(est, dev), 80.0))
    ref = O.metrics_batch(gt, est)
    assert np.max(rel_err(vals, ref, 1e-6)) < 2e-5, (vals, ref)
    # the class-based path of the product gives the same numbers
    mets = M.default_metrics()
    g = torch.clamp(to_dev(gt, dev), 0.0, 80.0)
    e = torch.clamp(to_dev(est, dev), 0.001, 80.0)
    for m in mets:
        m.update_state(g, e)
    cls = np.array([float(m.result()) for m in mets])
    assert np.max(rel_err(vals, cls, 1e-6)) < 2e-5


@pytest.mark.parametrize("b,h,w,cin,cout,slope,stride",
                         [(1, 16, 32, 16, 32, 0.1, 1), (2, 19, 37, 122, 128, 0.1, 1), (1, 24, 40, 128, 96, 0.1, 1),
                          (1, 9, 17, 64, 5, 1.0, 1), (1, 33, 20, 238, 64, 0.1, 1), (1, 8, 16, 470, 128, 0.1, 1),
                          (1, 40, 48, 32, 16, 0.1, 1), (2, 24, 40, 3, 16, 1.0, 1),
                          (1, 32, 48, 16, 16, 0.1, 2), (2, 19, 37, 64, 64, 0.1, 2), (1, 18, 21, 96, 96, 0.1, 2),
                          (1, 12, 40, 128, 192, 0.1, 2), (1, 7, 10, 128, 128, 0.1, 2)])
def test_mfma_conv3x3_bias_act(M, dev, b, h, w, cin, cout, slope, stride):
    """Hand-written fp32-MFMA 3x3 conv + bias + leaky_relu vs the oracle's conv2d_same (tolerance:
    1e-5 of the output scale -- different, but fixed, summation order) incl. ragged tiles, K and N
    padding, and run-to-run determinism."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    wp, cpad = nops.pack_conv_weights(k)
    xd, wd, bd = to_dev(x, dev), to_dev(wp, dev), to_dev(bias, dev)
    got = nops.conv3x3_bias_act(xd, wd, bd, cout, cpad, slope, stride=stride)
    ref = O.conv2d_same(x, k, bias, stride)
    ref = np.where(ref > 0, ref, ref * F(slope)).astype(F)
    err = np.max(np.abs(npy(got) - ref))
    assert err < 1e-5 * max(1.0, np.abs(ref).max()), err
    again = nops.conv3x3_bias_act(xd, wd, bd, cout, cpad, slope, stride=stride)
    assert torch.equal(got, again)


# ------------------------------------------------------------------ Winograd convolutions, fused tail, encoder head
@pytest.mark.parametrize("b,h,w,cin,cout,slope", [
    (2, 24, 40, 64, 128, 0.1),        # whole tiles
    (1, 37, 53, 122, 96, 0.1),        # ragged tiles, K padding (122 = 7 chunks + 10), one N-tile per workgroup
    (1, 17, 23, 32, 48, 1.0),         # N padding (48 -> 64), no activation
    (2, 16, 16, 16, 32, 0.1),         # a single chunk
    (1, 33, 70, 96, 64, 0.1),
])
def test_winograd_conv_kernel1(M, dev, b, h, w, cin, cout, slope):
    """Winograd F(2x2,3x3) convolution (16x8 tile, 16-channel chunks) vs the oracle's direct conv2d_same.  Tolerance 1e-5 of
    the output scale, the same as for the direct MFMA kernel (measured difference ~2e-6); deterministic."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(cin * 11 + cout)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    wu, cpad = nops.pack_conv_weights_winograd(k, chunk=16)
    xd, wd, bd = to_dev(x, dev), to_dev(wu, dev), to_dev(bias, dev)
    got = nops.conv3x3_wino_bias_act(xd, wd, bd, cout, cpad, slope)
    ref = O.conv2d_same(x, k, bias, 1)
    ref = np.where(ref > 0, ref, ref * F(slope)).astype(F)
    err = np.max(np.abs(npy(got) - ref))
    assert err < 1e-5 * max(1.0, np.abs(ref).max()), err
    assert torch.equal(got, nops.conv3x3_wino_bias_act(xd, wd, bd, cout, cpad, slope))


@pytest.mark.parametrize("b,h,w,cin,cout,slope", [
    (2, 32, 48, 64, 128, 0.1),
    (1, 37, 53, 124, 96, 0.1),        # ragged tiles, K padding (124 = 15 chunks + 4)
    (1, 19, 21, 32, 40, 1.0),         # N padding, no activation
    (2, 16, 16, 8, 32, 0.1),          # a single chunk
    (1, 50, 90, 96, 64, 0.1),
    (2, 100, 130, 40, 120, 0.1),      # >= 200 workgroups of 64 couts, Cin >= 32, Cin % 8 == 0: kernel 4 (ragged tiles,
    (1, 192, 320, 64, 64, 1.0),       # Cout < CoutPad); interior fast path + border tiles
])
def test_winograd_conv_kernel2(M, dev, b, h, w, cin, cout, slope):
    """Winograd kernel 2 (16x16 tile, 8-channel chunks, two M-tiles per wave) and kernel 4 (the same arithmetic on 512-thread
    workgroups, chosen by m4d_conv3x3_wino2_bias_act when the grid is large enough) vs the oracle; same tolerance."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(cin * 13 + cout)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    wu, cpad = nops.pack_conv_weights_winograd(k, chunk=8)
    xd, wd, bd = to_dev(x, dev), to_dev(wu, dev), to_dev(bias, dev)
    got = nops.conv3x3_wino2_bias_act(xd, wd, bd, cout, cpad, slope)
    ref = O.conv2d_same(x, k, bias, 1)
    ref = np.where(ref > 0, ref, ref * F(slope)).astype(F)
    err = np.max(np.abs(npy(got) - ref))
    assert err < 1e-5 * max(1.0, np.abs(ref).max()), err
    assert torch.equal(got, nops.conv3x3_wino2_bias_act(xd, wd, bd, cout, cpad, slope))


@pytest.mark.parametrize("b,h,w,cin,cout,slope", [
    (2, 32, 48, 64, 128, 0.1),
    (1, 37, 53, 112, 96, 0.1),        # ragged tiles; 96 output channels: a full 64-cout unit and a HALF unit (N-tile 0 only) per tile
    (2, 40, 56, 64, 32, 0.1),         # the refiner's 64 -> 32 layer: one half unit per tile
    (1, 33, 47, 16, 24, 0.1),         # a half unit with a single K chunk and a partly filled N-tile (guarded stores)
    (1, 19, 21, 32, 40, 1.0),         # N padding (40 -> 64), no activation, a map smaller than two tiles
    (2, 16, 16, 16, 64, 0.1),         # a single 16-channel chunk
    (1, 50, 90, 96, 64, 0.1),
    (2, 100, 130, 48, 120, 0.1),      # Cout % 4 == 0 but not a multiple of 64: the guarded store path on the last quad
    (1, 192, 320, 64, 66, 1.0),       # Cout % 4 != 0: scalar stores
])
def test_winograd_conv_bf16_split(M, dev, b, h, w, cin, cout, slope):
    """m4d_conv3x3_wino6_bias_act (Winograd F(2x2,3x3), float32 operands split exactly into three bf16 terms, six bf16 MFMA
    products, float32 accumulation) vs the oracle's convolution: the float32 tolerance of the fp32-MFMA Winograd kernels,
    AND no further from the float64 result than the float32 oracle itself (the claim that lets it stand in for them)."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(cin * 17 + cout)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    wu, cpad = nops.pack_conv_weights_wino6(k)
    assert cpad % 64 == 0 and wu.dtype == np.uint16
    xd, wd, bd = to_dev(x, dev), torch.from_numpy(wu.view(np.int16)).to(dev), to_dev(bias, dev)
    got = nops.conv3x3_wino6_bias_act(xd, wd, bd, cout, cpad, slope)
    act = lambda r: np.where(r > 0, r, r * r.dtype.type(slope))
    ref32 = act(O.conv2d_same(x, k, bias, 1)).astype(F)
    with O.float64_reference():
        ref64 = act(O.conv2d_same(x.astype(np.float64), k.astype(np.float64), bias.astype(np.float64), 1))
    assert ref64.dtype == np.float64
    scale = max(1.0, np.abs(ref32).max())
    assert np.max(np.abs(npy(got) - ref32)) < 1e-5 * scale
    e_gpu, e_oracle = np.abs(npy(got).astype(np.float64) - ref64), np.abs(ref32.astype(np.float64) - ref64)
    print(f"bf16-split Winograd {cin}->{cout}: mean |error| to float64: GPU {e_gpu.mean():.3e}, float32 oracle {e_oracle.mean():.3e}")
    # Winograd's transforms add 1.5-1.8x the rounding of the oracle's direct float32 sum; the fp32-MFMA Winograd kernel is the bar
    assert e_gpu.mean() <= 2.0 * e_oracle.mean() and e_gpu.max() <= 3.0 * e_oracle.max()
    if cin % 4 == 0:
        wu8, cpad8 = nops.pack_conv_weights_winograd(k, chunk=8)
        f32k = nops.conv3x3_wino2_bias_act(xd, to_dev(wu8, dev), bd, cout, cpad8, slope)
        e_f32 = np.abs(npy(f32k).astype(np.float64) - ref64)
        print(f"    fp32-MFMA Winograd kernel: {e_f32.mean():.3e}")
        assert e_gpu.mean() <= 1.02 * e_f32.mean()
    assert torch.equal(got, nops.conv3x3_wino6_bias_act(xd, wd, bd, cout, cpad, slope))          # deterministic


@pytest.mark.parametrize("b,h,w,cin,cout,slope", [
    (1, 192, 640, 128, 128, 0.1),     # the level-1 layer: 960 units on 256 workgroups (3-4 per workgroup, 8 K chunks)
    (1, 192, 640, 64, 128, 0.1),      # 4 chunks
    (1, 192, 640, 128, 96, 0.1),      # a second cout group of 32 channels: half units (N-tile 0 only)
    (1, 192, 640, 96, 64, 0.1),       # one cout group per tile: 480 units, 6 chunks
    (4, 100, 130, 48, 120, 0.1),      # odd chunk count (the raw-buffer parity flips per unit), ragged tiles, unit ranges crossing images
    (3, 100, 130, 32, 192, 1.0),      # two chunks (the shortest unit: both raw prefetches are the next unit's), three cout groups
    (2, 70, 150, 240, 66, 0.1),       # 15 chunks, Cout % 4 != 0: scalar stores
    (1, 40, 40, 64, 64, 0.1),         # 9 units: fewer than CUs, one unit per workgroup
    (9, 33, 17, 80, 70, 0.1),         # many small images
    # >= 1024 units: TEAM mode (the cout groups of a tile on n_groups workgroups of one XCD, XCD-banded tile runs)
    (2, 192, 640, 64, 128, 0.1),      # 1920 units, teams of two
    (9, 100, 130, 48, 120, 0.1),      # 1134 units, odd chunk count, ragged tiles, tile runs crossing images
    (6, 100, 130, 32, 192, 1.0),      # teams of three: 10 teams per XCD, two workgroups per XCD idle
    (3, 192, 640, 96, 64, 0.1),       # teams of one (a single cout group): XCD-banded contiguous runs
    # HALF units (a last cout group of <= 32 channels runs N-tile 0 only) in team mode: the members swap groups tile by tile
    (3, 192, 640, 128, 96, 0.1),      # the refiner's 128 -> 96 layer, 2880 units
    (6, 100, 130, 32, 160, 1.0),      # teams of three, two full groups + a half one, two chunks, ragged tiles
    (5, 100, 130, 48, 66, 0.1),       # the half group holds two channels: scalar stores
])
def test_persistent_winograd_is_bitwise_the_one_tile_kernel(M, dev, b, h, w, cin, cout, slope):
    """m4d_wino6p.hip (persistent workgroups walking (tile, cout group) units, the K loop's DMA stream continuing across
    unit boundaries, four-pass epilogue in two ring slots) against m4d_wino6.hip (one workgroup per unit): the same float32
    bits -- same products, same accumulation order, same association in the output transform -- through the explicit
    kernel argument of m4d_conv3x3_wino6_bias_act_k and through the default dispatch; repeated launches agree.  Also the
    short-range form (kernel = 16 + n: workgroups of n consecutive units, placed by the dispatcher; round 6)."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(b * 1000 + h + cin + cout)
    x = to_dev(rng.standard_normal([b, h, w, cin]).astype(F), dev)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = to_dev((0.1 * rng.standard_normal([cout])).astype(F), dev)
    wu6, cpad = nops.pack_conv_weights_wino6(k)
    wud = torch.from_numpy(wu6.view("int16")).to(dev)
    one = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, slope, kernel=1)
    per = [nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, slope, kernel=2) for _ in range(3)]
    # round 6: persistent workgroups of n consecutive units each (kernel = 16 + n), with and without the staggered first round
    per += [nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, slope, kernel=16 + n, stagger_us=us) for n, us in ((2, 0), (2, 9), (3, 0), (5, 13))]
    auto = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, slope)
    for o in per:
        ne = o.view(torch.int32) != one.view(torch.int32)
        assert not bool(ne.any()), f"{int(ne.sum())} of {one.numel()} elements differ; first at (b,y,x,c) {ne.nonzero()[0].tolist()}"
    assert torch.equal(auto, one)
    ref = O.leaky_relu(O.conv2d_same(npy(x), k, npy(bias), 1), slope) if slope != 1.0 else O.conv2d_same(npy(x), k, npy(bias), 1)
    assert np.max(np.abs(npy(per[0]) - ref)) < 1e-5 * max(1.0, np.abs(ref).max())


def test_winograd_staggered_first_round_changes_no_bit(M, dev):
    """m4d_conv3x3_wino6_bias_act_ks: with stagger_us > 0 the first 256 workgroups of a launch start in phases (a delay in front of
    the kernel body, so that the CUs do not free in lock step, csrc/m4d_wino6.hip) -- off, the default and extreme settings give
    the same bits on a multi-round grid (960 units), a single-round one (240) and a small one; bad arguments are refused."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(5)
    for (h, w, cin, cout) in ((192, 640, 32, 128), (96, 320, 64, 128), (48, 160, 32, 64)):
        x = to_dev(rng.standard_normal([1, h, w, cin]).astype(F), dev)
        k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
        bias = to_dev((0.1 * rng.standard_normal([cout])).astype(F), dev)
        wu6, cpad = nops.pack_conv_weights_wino6(k)
        wud = torch.from_numpy(wu6.view("int16")).to(dev)
        outs = [nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1, kernel=1, stagger_us=us, stagger_phases=ph)
                for us, ph in ((0, 16), (9, 16), (40, 32), (13, 8), (300, 1))]
        outs.append(nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1, kernel=2, stagger_us=9, stagger_phases=16))   # persistent: ignored
        for o in outs[1:]:
            assert torch.equal(o, outs[0])
        for us, ph in ((-1, 16), (9, 3), (9, 64), (2000, 16)):
            with pytest.raises(RuntimeError):
                nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1, kernel=1, stagger_us=us, stagger_phases=ph)


def test_winograd_stagger_is_per_call(M, dev):
    """ABI 6: the staggered first round is an ARGUMENT of the launch, not library state.  Two host threads launch
    the same layer at the same time on their own streams, one in lock step and one with a 200-us second phase (the kernel bounds its
    wait at ~250 us: a delay, never a hang): every launch of the first stays short, every launch of the second carries its own delay (with round 5's process-wide setter both threads saw
    whichever value was written last), and both get the same bits."""
    import threading
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(11)
    h, w, cin, cout = 48, 160, 32, 64                      # 30 workgroups: both threads' launches fit on the chip side by side
    x = to_dev(rng.standard_normal([1, h, w, cin]).astype(F), dev)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = to_dev((0.1 * rng.standard_normal([cout])).astype(F), dev)
    wu6, cpad = nops.pack_conv_weights_wino6(k)
    wud = torch.from_numpy(wu6.view("int16")).to(dev)
    torch.cuda.synchronize()
    res, start = {}, threading.Barrier(2)

    def worker(name, us, phases):
        st = torch.cuda.Stream()
        times, outs = [], []
        with torch.cuda.stream(st):
            for _ in range(3):                             # warm-up (module load, first-touch)
                nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1, kernel=1, stagger_us=us, stagger_phases=phases)
            st.synchronize()
            start.wait()
            torch.cuda._sleep(int(2.0e7))                  # the host queues the timed launches behind a GPU-side spin: an event pair
            for _ in range(24):                            # then brackets the kernel's execution, not the host's launch gaps
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                outs.append(nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1, kernel=1, stagger_us=us, stagger_phases=phases))
                e1.record(st)
                times.append((e0, e1))
            st.synchronize()
        res[name] = ([a.elapsed_time(b) * 1e3 for a, b in times], outs)

    th = [threading.Thread(target=worker, args=("lock_step", 0, 1)), threading.Thread(target=worker, args=("staggered", 400, 2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    fast, slow = res["lock_step"][0], res["staggered"][0]
    # (relative bounds: the launches of the two threads share the chip, so absolute durations move with the box)
    assert min(slow) >= 180.0, f"a staggered launch lost its delay: {sorted(slow)[:4]} us"
    assert float(np.median(fast)) + 100.0 < min(slow) and min(fast) < 100.0, \
        f"the lock-step thread's launches carry the other thread's delay: median {np.median(fast):.0f} / min {min(fast):.0f} us against {min(slow):.0f}"
    for o in res["lock_step"][1] + res["staggered"][1]:
        assert torch.equal(o, res["lock_step"][1][0])


@pytest.mark.parametrize("cin,cout", [(128, 128), (32, 128), (16, 128), (128, 96)])
def test_winograd_bf16_split_determinism_under_memory_pressure(M, dev, cin, cout):
    """Regression test of round 3's non-determinism: the level-1 refiner layer geometry at batch 32,
    200 launches each queued behind streaming HBM copy traffic on a side stream, against the quiet run, bit for bit.  The
    round-3 library differs in ~4 % of such launches (whole output tiles of 32 couts wrong): the prologue of
    conv3x3_wino6_kernel issued raw(1) before B(2), so the wait that closes position 0 of the first chunk left B(2)'s third
    piece in flight and position 1 read it from LDS covered by elapsed time only.  Cin = 32 / 16: two / one K chunk, where
    the K loop's surplus DMAs (chunks past the last) are in flight when the epilogue reuses the LDS.  Cout = 96: full and
    half units (N-tile 0 only) side by side, the persistent kernel's team members alternating between them."""
    from m4depth_amd import network_ops as nops
    from helpers import hbm_pressure
    torch.manual_seed(5)
    b, h, w = 32, 192, 640
    if cin != 128 or cout != 128:
        b = 8
    x = torch.randn(b, h, w, cin, device=dev)
    k = (torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5).numpy()
    bias = torch.randn(cout, device=dev) * 0.1
    wu, cpad = nops.pack_conv_weights_wino6(k)
    wd = torch.from_numpy(wu.view(np.int16)).to(dev)
    quiet = nops.conv3x3_wino6_bias_act(x, wd, bias, cout, cpad, 0.1, kernel=1).clone()
    torch.cuda.synchronize()
    load = hbm_pressure(dev)
    for kernel in ((1, 2) if cin >= 32 else (1,)):       # one workgroup per unit (m4d_wino6.hip), persistent (m4d_wino6p.hip)
        n_bad = torch.zeros((), dtype=torch.int64, device=dev)
        for it in range(200 if b == 32 else 60):
            if it % 4 == 0:
                load.queue(12)
            out = nops.conv3x3_wino6_bias_act(x, wd, bias, cout, cpad, 0.1, kernel=kernel)
            n_bad += (out.view(torch.int32) != quiet.view(torch.int32)).any().to(torch.int64)
        torch.cuda.synchronize()
        assert int(n_bad) == 0, f"kernel {kernel}: {int(n_bad)} launches under memory pressure differ from the quiet launch"


@pytest.mark.parametrize("b,h,w,cin,cout,slope", [(1, 6, 20, 472, 128, 0.1), (2, 12, 40, 240, 128, 0.1), (1, 24, 80, 128, 96, 0.1),
                                                  (1, 7, 9, 100, 40, 1.0), (3, 5, 33, 16, 32, 0.1), (1, 13, 17, 64, 5, 1.0)])
def test_small_map_conv_bf16_split(M, dev, b, h, w, cin, cout, slope):
    """m4d_conv3x3_small6_bias_act (the one-launch small-map convolution with float32 operands split into three bf16 terms) vs the
    oracle: the tolerance of the fp32-MFMA kernel, an error to the float64 result no larger than the fp32 kernel's, bitwise
    deterministic; ragged tiles, a partial last chunk (Cin = 100), output-channel padding, batch."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(cin * 3 + cout)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    wp6, cpad = nops.pack_conv_weights_small6(k)
    xd, bd = to_dev(x, dev), to_dev(bias, dev)
    wd = torch.from_numpy(wp6.view(np.int16)).to(dev)
    got = nops.conv3x3_small6_bias_act(xd, wd, bd, cout, cpad, slope)
    act = lambda r: np.where(r > 0, r, r * r.dtype.type(slope))
    ref32 = act(O.conv2d_same(x, k, bias, 1)).astype(F)
    with O.float64_reference():
        ref64 = act(O.conv2d_same(x.astype(np.float64), k.astype(np.float64), bias.astype(np.float64), 1))
    assert np.max(np.abs(npy(got) - ref32)) < 1e-5 * max(1.0, np.abs(ref32).max())
    wp, cpad32 = nops.pack_conv_weights(k)
    f32k = nops.conv3x3_small_bias_act(xd, to_dev(wp, dev), bd, cout, cpad32, slope)
    e6, e32 = np.abs(npy(got).astype(np.float64) - ref64).mean(), np.abs(npy(f32k).astype(np.float64) - ref64).mean()
    print(f"small-map conv {cin}->{cout}: mean |error| to float64: bf16 split {e6:.3e}, fp32 MFMA {e32:.3e}")
    assert e6 <= 1.05 * e32
    assert torch.equal(got, nops.conv3x3_small6_bias_act(xd, wd, bd, cout, cpad, slope))


@pytest.mark.parametrize("b,h,w,cin,cout,slope", [(1, 6, 20, 472, 128, 0.1), (1, 12, 40, 240, 128, 0.1), (1, 24, 80, 128, 96, 0.1),
                                                  (1, 48, 160, 96, 64, 0.1), (2, 7, 9, 100, 40, 1.0), (3, 5, 33, 16, 32, 0.1),
                                                  (1, 13, 17, 64, 32, 0.1), (1, 9, 11, 32, 5, 1.0)])
def test_latency_conv_vs_oracle(M, dev, b, h, w, cin, cout, slope):
    """m4d_conv3x3_lat (round 5: the latency-first small-map convolution, csrc/m4d_convlat.hip) vs the oracle, every variant:
    1 / 2 / 4 M-tiles per wave x 1 / 2 / 4 K sub-slices per workgroup x 1-4 K slices over workgroups (partial slabs finished by
    m4d_partial_finish).  Tolerance of the other float32 convolution kernels; error to float64 no larger than the bf16-split
    small-map kernel's; variants with the same (kw, s_out) are bitwise equal whatever the M-tile count; the chosen default
    configuration is among them; ragged tiles, a partial last chunk (Cin = 100), Cout not a multiple of 32, batch."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(cin * 5 + cout)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    xd, bd = to_dev(x, dev), to_dev(bias, dev)
    wd = torch.from_numpy(nops.pack_conv_weights_lat(k).view(np.int16)).to(dev)
    act = lambda r: np.where(r > 0, r, r * r.dtype.type(slope))
    ref32 = act(O.conv2d_same(x, k, bias, 1)).astype(F)
    with O.float64_reference():
        ref64 = act(O.conv2d_same(x.astype(np.float64), k.astype(np.float64), bias.astype(np.float64), 1))
    wp6, cpad = nops.pack_conv_weights_small6(k)
    small6 = nops.conv3x3_small6_bias_act(xd, torch.from_numpy(wp6.view(np.int16)).to(dev), bd, cout, cpad, slope)
    e_small6 = np.abs(npy(small6).astype(np.float64) - ref64).mean()
    n_chunks = -(-cin // 16)
    same_order = {}
    tried = 0
    for kw in (1, 2, 4):
        for s_out in (1, 2, 3, 4):
            if s_out > n_chunks or (s_out - 1) * (-(-n_chunks // s_out)) >= n_chunks:
                continue
            for mt in (1, 2, 4):
                if 2 * kw * {1: 60, 2: 100, 4: 180}[mt] * 96 > 160 * 1024:
                    continue
                out = nops.conv3x3_lat(xd, wd, bd, cout, slope, config=(mt, kw, s_out))
                if s_out > 1:
                    assert isinstance(out, nops.PartialAct) and out.slabs.shape == (s_out, b, h, w, cout)
                    if cout % 4 != 0:
                        continue                                   # m4d_partial_finish wants whole channel quads
                    out = out.dense()
                got = npy(out)
                tried += 1
                assert np.max(np.abs(got - ref32)) < 1e-5 * max(1.0, np.abs(ref32).max()), (mt, kw, s_out)
                e = np.abs(got.astype(np.float64) - ref64).mean()
                # (one wave adding all of a long K in one accumulator -- kw = s_out = 1 on 472 channels -- rounds more often
                # than the four interleaved chains of the small-map kernel: within 2x; the dispatched configurations split K
                # at least as finely as that kernel and are held to its error below)
                assert e <= 2.0 * e_small6 + 1e-12, (mt, kw, s_out, e, e_small6)
                key = (kw, s_out)
                if key in same_order:
                    assert torch.equal(out, same_order[key]), f"mt {mt} differs bitwise from another M-tile count at {key}"
                else:
                    same_order[key] = out
    assert tried >= 6
    mw = nops.conv3x3_lat(xd, wd, bd, cout, slope, config=(8, 1, 1))       # "M over waves": the kw = 1, s_out = 1 order
    assert torch.equal(mw, same_order[(1, 1)])
    default = nops.conv3x3_lat(xd, wd, bd, cout, slope, final=True)
    cfg = nops.lat_config(b, h, w, cin, cout, final=True)
    assert cfg[2] == 1 and torch.equal(default, same_order[(cfg[1], 1)])
    assert torch.equal(default, nops.conv3x3_lat(xd, wd, bd, cout, slope, final=True))     # deterministic


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 48, 160, 96, 96), (2, 24, 80, 128, 128), (1, 12, 40, 192, 192), (1, 96, 320, 64, 64),
                                            (1, 13, 17, 32, 40), (3, 7, 10, 100, 20), (1, 8, 9, 16, 32)])
def test_latency_conv_stride2_vs_oracle(M, dev, b, h, w, cin, cout):
    """m4d_conv3x3s_lat at stride 2 (the coarse stride-2 encoder layers, m4depth_network.py:66-72): TF 'SAME' padding on even
    and odd sizes (pad before 0 / 1), every (mt, kw, s_out) that fits the LDS, against the oracle and the fp32-MFMA one-launch
    kernel it replaces; partial slabs consumed by a following stride-1 call bit for bit as the finished tensor."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    xd, bd = to_dev(x, dev), to_dev(bias, dev)
    wd = torch.from_numpy(nops.pack_conv_weights_lat(k).view(np.int16)).to(dev)
    r = O.conv2d_same(x, k, bias, 2)
    ref32 = np.where(r > 0, r, r * F(0.1)).astype(F)
    oh, ow = -(-h // 2), -(-w // 2)
    assert ref32.shape == (b, oh, ow, cout)
    n_chunks = -(-cin // 16)
    tried = 0
    same_order = {}
    for kw in (1, 2, 4):
        for s_out in (1, 2, 4):
            if s_out > n_chunks or (s_out - 1) * (-(-n_chunks // s_out)) >= n_chunks:
                continue
            for mt in (1, 2, 4):
                if 2 * kw * nops._lat_halo_pixels(mt, 2) * 96 > 160 * 1024:
                    continue
                out = nops.conv3x3_lat(xd, wd, bd, cout, 0.1, config=(mt, kw, s_out), stride=2)
                if s_out > 1:
                    out = out.dense()
                tried += 1
                assert np.max(np.abs(npy(out) - ref32)) < 1e-5 * max(1.0, np.abs(ref32).max()), (mt, kw, s_out)
                key = (kw, s_out)
                if key in same_order:
                    assert torch.equal(out, same_order[key]), (mt, kw, s_out)
                else:
                    same_order[key] = out
    assert tried >= 4
    for s_out in (1, 2):                                # "M over waves" (mt code 8): K unsplit inside the workgroup = the kw 1 order
        if (1, s_out) in same_order:
            out = nops.conv3x3_lat(xd, wd, bd, cout, 0.1, config=(8, 1, s_out), stride=2)
            assert torch.equal(out.dense() if s_out > 1 else out, same_order[(1, s_out)]), ("mw", s_out)
    default = nops.conv3x3_lat(xd, wd, bd, cout, 0.1, final=True, stride=2)
    cfg = nops.lat_config(b, h, w, cin, cout, True, 2)
    assert torch.equal(default, same_order[(cfg[1], 1)])
    if n_chunks >= 2 and cout % 4 == 0 and cout >= 16:
        # stride-2 partial slabs -> a stride-1 consumer
        k2 = (rng.standard_normal([3, 3, cout, 32]) * np.sqrt(2.0 / (9 * cout))).astype(F)
        w2 = torch.from_numpy(nops.pack_conv_weights_lat(k2).view(np.int16)).to(dev)
        b2 = to_dev(np.zeros(32, F), dev)
        p = nops.conv3x3_lat(xd, wd, bd, cout, 0.1, config=(1, 1, 2), stride=2)
        assert isinstance(p, nops.PartialAct)
        assert torch.equal(nops.conv3x3_lat(p, w2, b2, 32, 0.1, final=True), nops.conv3x3_lat(p.dense(), w2, b2, 32, 0.1, final=True))


def test_latency_conv_chain_finishes_partial_sums_while_staging(M, dev):
    """A chain of m4d_conv3x3_lat calls hands K-slice partial sums from layer to layer (``PartialAct``): the consumer adds the
    slabs in slab order, the producer's bias and leaky_relu while it stages its halo -- bit for bit what it computes from the
    finished tensor of m4d_partial_finish -- for 2, 3 and 4 slabs; the whole refiner prefix of a coarse level (472 -> 128 ->
    128 -> 96 -> 64 -> 32 at 6x20) against the oracle."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(77)
    b, h, w = 1, 6, 20
    chans = [472, 128, 128, 96, 64, 32]
    x = rng.standard_normal([b, h, w, chans[0]]).astype(F)
    ks = [(rng.standard_normal([3, 3, ci, co]) * np.sqrt(2.0 / (9 * ci))).astype(F) for ci, co in zip(chans[:-1], chans[1:])]
    bs = [(0.1 * rng.standard_normal([co])).astype(F) for co in chans[1:]]
    wds = [torch.from_numpy(nops.pack_conv_weights_lat(k).view(np.int16)).to(dev) for k in ks]
    bds = [to_dev(bb, dev) for bb in bs]
    xd = to_dev(x, dev)
    for s_out in (2, 3, 4):
        p = nops.conv3x3_lat(xd, wds[0], bds[0], chans[1], 0.1, config=(1, 4, s_out))
        assert isinstance(p, nops.PartialAct)
        for cfg in ((1, 1, 1), (1, 4, 1), (2, 2, 2)):
            a = nops.conv3x3_lat(p, wds[1], bds[1], chans[2], 0.1, config=cfg)
            c = nops.conv3x3_lat(p.dense(), wds[1], bds[1], chans[2], 0.1, config=cfg)
            if cfg[2] > 1:
                a, c = a.dense(), c.dense()
            assert torch.equal(a, c), (s_out, cfg)
    # default configurations, layer after layer
    cur, ref = xd, x
    for i in range(5):
        cur = nops.conv3x3_lat(cur, wds[i], bds[i], chans[i + 1], 0.1, final=(i == 4))
        r = O.conv2d_same(ref, ks[i], bs[i], 1)
        ref = np.where(r > 0, r, r * F(0.1)).astype(F)
    assert isinstance(cur, torch.Tensor)
    assert np.max(np.abs(npy(cur) - ref)) < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("kernel", ["f32", "bf16x3"])
@pytest.mark.parametrize("b,h,w,quat", [(2, 24, 40, True), (1, 37, 53, False), (1, 6, 20, True), (3, 96, 320, True), (1, 10, 14, True),
                                        (1, 11, 15, False), (1, 3, 5, True)])
def test_fused_refiner_tail(M, dev, b, h, w, quat, kernel):
    """conv(32->16)+lrelu, conv(16->5) and the level tail in one kernel -- on the fp32 matrix cores (m4d_tail.hip) and in the
    bf16-split arithmetic with persistent workgroups (m4d_tail6.hip) -- vs the oracle's two convolutions + the oracle's
    exp/clip/parallax2depth: refiner output within 1e-5, parallax 3e-5 relative (exp of a value known to 1e-5), depth =
    parallax2depth(parallax) bit for bit; ragged tiles, maps smaller than one tile, more tiles than workgroups, batch."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(h * w)
    x = np.maximum(rng.standard_normal([b, h, w, 32]), -0.3).astype(F)
    k6 = (rng.standard_normal([3, 3, 32, 16]) * np.sqrt(2.0 / (9 * 32))).astype(F)
    k7 = (rng.standard_normal([3, 3, 16, 5]) * np.sqrt(1.0 / (9 * 16))).astype(F)
    b6 = (0.1 * rng.standard_normal([16])).astype(F)
    b7 = (0.1 * rng.standard_normal([5])).astype(F)
    rot, trans = motion_np(rng, b, quat=quat)
    cam = camera_np(b, h, w)
    scale = F(0.5)
    mid = O.conv2d_same(x, k6, b6, 1)
    mid = np.where(mid > 0, mid, mid * F(0.1)).astype(F)
    out5 = O.conv2d_same(mid, k7, b7, 1)
    para_ref = (np.exp(np.clip(out5[..., :1], F(-7), F(7))) / scale).astype(F)
    state = torch.zeros((b, h, w, 1), device=dev)
    if kernel == "f32":
        w6, w7 = nops.pack_refiner_tail_weights(k6, k7)
        call = lambda: nops.refiner_tail(to_dev(x, dev), to_dev(w6, dev), to_dev(b6, dev), to_dev(w7, dev), to_dev(b7, dev),
                                         to_dev(rot, dev), to_dev(trans, dev), to_dev(cam, dev), float(scale), depth_state=state)
    else:
        w6, w7 = nops.pack_refiner_tail_weights6(k6, k7)
        w6d, w7d = torch.from_numpy(w6.view(np.int16)).to(dev), torch.from_numpy(w7.view(np.int16)).to(dev)
        call = lambda: nops.refiner_tail6(to_dev(x, dev), w6d, to_dev(b6, dev), w7d, to_dev(b7, dev),
                                          to_dev(rot, dev), to_dev(trans, dev), to_dev(cam, dev), float(scale), depth_state=state)
    para, depth, other = call()
    assert np.max(np.abs(npy(other) - out5[..., 1:])) < 1e-5 * max(1.0, np.abs(out5).max())
    assert rel_err(npy(para), para_ref).max() < 3e-5                      # exp of a value known to 1e-5
    depth_from_gpu_para = O.parallax2depth(npy(para), rot, trans, cam)    # the converter itself is bit-exact elsewhere
    assert_bits_equal(npy(depth), depth_from_gpu_para, "depth = parallax2depth(parallax)")
    assert torch.equal(state, depth)
    para2, depth2, other2 = call()                                        # deterministic
    assert torch.equal(para, para2) and torch.equal(depth, depth2) and torch.equal(other, other2)
    with O.float64_reference():                                           # error to float64: the class of the float32 oracle
        mid64 = O.conv2d_same(x.astype(np.float64), k6.astype(np.float64), b6.astype(np.float64), 1)
        mid64 = np.where(mid64 > 0, mid64, mid64 * 0.1)
        out64 = O.conv2d_same(mid64, k7.astype(np.float64), b7.astype(np.float64), 1)
    e_gpu = np.abs(npy(other).astype(np.float64) - out64[..., 1:]).mean()
    e_oracle = np.abs(out5[..., 1:].astype(np.float64) - out64[..., 1:]).mean()
    print(f"tail {kernel} {b}x{h}x{w}: mean |error| to float64 {e_gpu:.3e} (float32 oracle {e_oracle:.3e})")
    assert e_gpu <= 3.0 * e_oracle + 1e-9


def _trained_legacy_encoder_weights():
    """The trained first two encoder levels of the reference's legacy model, read from the TensorFlow-written bundle
    head in tests/golden/tf_legacy (same layer shapes as FeaturePyramid, m4depth_network.py:59-74)."""
    import os
    from m4depth_amd import tf_checkpoint as TC
    d = os.path.join(os.path.dirname(__file__), "golden", "tf_legacy")
    r = TC.CheckpointReader(os.path.join(d, "features"), data_path=os.path.join(d, "features.data-%05d-of-%05d"))
    return {(l, c, p): r.tensor(f"feature_pyramid/layer_{l}/conv2d_{c}/{p}") for l in (1, 2) for c in (1, 2) for p in ("kernel", "bias")}


@pytest.mark.parametrize("weights", ["random", "trained"])
def test_fused_encoder_head(M, dev, weights):
    """Direct 3->16 convolution + bias + DINL statistics, and the stride-2 convolution with the DINL apply fused into its
    staging, vs the oracle's conv -> domain_normalization -> leaky_relu -> conv(stride 2) -> leaky_relu; with random
    weights and with the trained weights of the reference's legacy encoder."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(77)
    b, h, w = 2, 36, 52
    img = rng.random([b, h, w, 3]).astype(F)
    k1 = (rng.standard_normal([3, 3, 3, 16]) * np.sqrt(2.0 / 27)).astype(F)
    b1 = (0.1 * rng.standard_normal([16])).astype(F)
    k2 = (rng.standard_normal([3, 3, 16, 16]) * np.sqrt(2.0 / 144)).astype(F)
    b2 = (0.1 * rng.standard_normal([16])).astype(F)
    if weights == "trained":
        tw = _trained_legacy_encoder_weights()
        k1, b1, k2, b2 = tw[1, 1, "kernel"], tw[1, 1, "bias"], tw[1, 2, "kernel"], tw[1, 2, "bias"]
    sc = (1.0 + 0.1 * rng.standard_normal([16])).astype(F)
    bs = (0.1 * rng.standard_normal([16])).astype(F)
    t = O.conv2d_same(img, k1, b1, 1)
    t = O.leaky_relu(O.domain_normalization(t, sc, bs), 0.1)
    ref = O.leaky_relu(O.conv2d_same(t, k2, b2, 2), 0.1)
    wp2, cpad2 = nops.pack_conv_weights(k2)
    got = nops.encoder_head(to_dev(img, dev), to_dev(k1.reshape(27, 16).copy(), dev), to_dev(b1, dev), to_dev(sc, dev),
                            to_dev(bs, dev), to_dev(wp2, dev), to_dev(b2, dev), 16, cpad2, 0.1)
    err = np.max(np.abs(npy(got) - ref))
    assert err < 2e-5 * max(1.0, np.abs(ref).max()), err


@pytest.mark.parametrize("weights,b,h,w", [("random", 2, 36, 52), ("trained", 2, 36, 52), ("random", 1, 37, 53),
                                           ("random", 3, 16, 130), ("random", 1, 64, 272)])
def test_encoder_level0_without_intermediate(M, dev, weights, b, h, w):
    """m4d_enc_level0_fwd: conv 3->16 recomputed on 16x16x4 MFMAs in three passes (sums, squared deviations, normalise +
    stride-2 convolution), never writing the [b,h,w,16] map -- vs the oracle's conv -> domain_normalization -> leaky_relu ->
    conv(stride 2) -> leaky_relu; even and odd sizes (TF 'SAME' pads differently), tiles ragged in both directions, several
    statistics tiles per workgroup row, and the frames of a sequence batch read in place (FrameStack)."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(h * 7 + w)
    img = rng.random([b, h, w, 3]).astype(F)
    k1 = (rng.standard_normal([3, 3, 3, 16]) * np.sqrt(2.0 / 27)).astype(F)
    b1 = (0.1 * rng.standard_normal([16])).astype(F)
    k2 = (rng.standard_normal([3, 3, 16, 16]) * np.sqrt(2.0 / 144)).astype(F)
    b2 = (0.1 * rng.standard_normal([16])).astype(F)
    if weights == "trained":
        tw = _trained_legacy_encoder_weights()
        k1, b1, k2, b2 = tw[1, 1, "kernel"], tw[1, 1, "bias"], tw[1, 2, "kernel"], tw[1, 2, "bias"]
    sc = (1.0 + 0.1 * rng.standard_normal([16])).astype(F)
    bs = (0.1 * rng.standard_normal([16])).astype(F)
    t = O.conv2d_same(img, k1, b1, 1)
    t = O.leaky_relu(O.domain_normalization(t, sc, bs), 0.1)
    ref = O.leaky_relu(O.conv2d_same(t, k2, b2, 2), 0.1)
    args = (to_dev(k1.copy(), dev), to_dev(b1, dev), to_dev(sc, dev), to_dev(bs, dev), to_dev(k2.copy(), dev), to_dev(b2, dev), 0.1)
    got = nops.encoder_level0(to_dev(img, dev), *args)
    assert got.shape == ref.shape
    err = np.max(np.abs(npy(got) - ref))
    assert err < 2e-5 * max(1.0, np.abs(ref).max()), err
    assert torch.equal(got, nops.encoder_level0(to_dev(img, dev), *args))          # deterministic
    if b >= 2:                                       # [bsz=1, T=b] sequence batch read in place = the dense batch, bit for bit
        seq = to_dev(img[None], dev)                 # [1, T, h, w, 3]
        assert torch.equal(got, nops.encoder_level0(nops.FrameStack(seq), *args))


@pytest.mark.parametrize("b,h,w,offset,spread", [(2, 96, 272, 0.0, 1.0), (1, 384, 1280, 0.0, 1.0), (3, 37, 53, 0.9, 0.01), (1, 130, 260, 50.0, 0.5)])
def test_encoder_level0_statistics_single_pass(M, dev, b, h, w, offset, spread):
    """m4d_enc_level0_stats, round 6: mean and variance of the (never written) 3 -> 16 convolution output from ONE pass over the
    image -- shifted moments sum (y - K), sum (y - K)^2 with K = the convolution at the centre pixel, merged in double -- instead
    of four launches (RGB totals, analytic mean, squared deviations, finalisation).  Against the float64 moments of the oracle's
    convolution, also on images whose mean dwarfs their spread (where an unshifted E[y^2] - E[y]^2 loses every digit)."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(h + w)
    img = (offset + spread * rng.random([b, h, w, 3])).astype(F)
    k1 = (rng.standard_normal([3, 3, 3, 16]) * np.sqrt(2.0 / 27)).astype(F)
    b1 = (0.1 * rng.standard_normal([16])).astype(F)
    y = O.conv2d_same(img, k1, b1, 1).astype(np.float64)
    mean_ref, var_ref = y.mean(axis=(1, 2)), y.var(axis=(1, 2))
    mean, var = nops.encoder_level0_stats(to_dev(img, dev), to_dev(k1.copy(), dev), to_dev(b1, dev))
    m, v = npy(mean).astype(np.float64), npy(var).astype(np.float64)
    assert np.max(np.abs(m - mean_ref) / (np.abs(mean_ref) + np.sqrt(var_ref))) < 2e-6
    assert np.max(np.abs(v - var_ref) / var_ref) < 2e-5, np.max(np.abs(v - var_ref) / var_ref)
    m2, v2 = nops.encoder_level0_stats(to_dev(img, dev), to_dev(k1.copy(), dev), to_dev(b1, dev))
    assert torch.equal(mean, m2) and torch.equal(var, v2)                      # deterministic


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 48, 160, 96, 96), (1, 24, 80, 128, 128), (2, 12, 40, 192, 192), (1, 13, 21, 32, 40),
                                            (1, 9, 10, 100, 64)])
def test_small_map_conv_stride2(M, dev, b, h, w, cin, cout):
    """The one-launch small-map kernel at stride 2 (the coarse stride-2 layers of the encoder) vs the oracle: TF 'SAME' on even
    and odd sizes, ragged tiles, a partial last chunk, output-channel padding; and the stride-1 entry unchanged."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(cin + cout + h)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    wp, cpad = nops.pack_conv_weights(k)
    xd, wd, bd = to_dev(x, dev), to_dev(wp, dev), to_dev(bias, dev)
    for stride in (2, 1):
        got = nops.conv3x3_small_bias_act(xd, wd, bd, cout, cpad, 0.1, stride=stride)
        ref = O.leaky_relu(O.conv2d_same(x, k, bias, stride), 0.1)
        assert got.shape == ref.shape
        assert np.max(np.abs(npy(got) - ref)) < 1e-5 * max(1.0, np.abs(ref).max()), stride
        assert torch.equal(got, nops.conv3x3_small_bias_act(xd, wd, bd, cout, cpad, 0.1, stride=stride))
        # the split-K kernel sums the same products in another order: equal to float32 rounding, not bit for bit
        other = nops.conv3x3_bias_act(xd, wd, bd, cout, cpad, 0.1, stride=stride)
        assert torch.max(torch.abs(got - other)).item() < 2e-6 * max(1.0, np.abs(ref).max()), stride


def test_encoder_level_2_with_trained_weights(M, dev):
    """16->32 stride 1 and 32->32 stride 2 (TF SAME) through the MFMA convolution with the trained weights of the
    reference's legacy encoder (tests/golden/tf_legacy) vs the oracle."""
    from m4depth_amd import network_ops as nops
    tw = _trained_legacy_encoder_weights()
    rng = np.random.default_rng(5)
    x = rng.standard_normal([2, 45, 70, 16]).astype(F)
    ref = O.leaky_relu(O.conv2d_same(x, tw[2, 1, "kernel"], tw[2, 1, "bias"], 1), 0.1)
    ref = O.leaky_relu(O.conv2d_same(ref, tw[2, 2, "kernel"], tw[2, 2, "bias"], 2), 0.1)
    t = to_dev(x, dev)
    for c, stride in ((1, 1), (2, 2)):
        wp, cpad = nops.pack_conv_weights(tw[2, c, "kernel"])
        t = nops.conv3x3_bias_act(t, to_dev(wp, dev), to_dev(tw[2, c, "bias"], dev), tw[2, c, "kernel"].shape[3], cpad, 0.1, stride)
    err = np.max(np.abs(npy(t) - ref))
    assert t.shape == (2, 23, 35, 32) and err < 1e-5 * max(1.0, np.abs(ref).max()), err


def test_winograd_kernel_4_is_bitwise_kernel_2(dev, tmp_path):
    """m4d_conv3x3_wino2_bias_act picks kernel 2 or kernel 4 by grid size; both must give the same bits (same products,
    same order).  The choice is read once per process (M4D_WINO_VARIANT), hence two subprocesses."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "wino_variant_check.py")
    outs = []
    for variant in ("2", "4"):
        env = dict(os.environ, M4D_WINO_VARIANT=variant, M4D_WINO4_MIN_WG="0")
        out = str(tmp_path / f"v{variant}.pt")
        subprocess.run([sys.executable, tool, "--save", out], check=True, env=env, timeout=600, capture_output=True)
        outs.append(out)
    res = subprocess.run([sys.executable, tool, "--compare", *outs], check=True, timeout=600, capture_output=True, text=True).stdout
    lines = [l for l in res.strip().splitlines() if l]
    assert len(lines) >= 5 and all(l.endswith("bit-identical") for l in lines), res
    # ... and both against the oracle's direct convolution at these geometries (Winograd differs by float32 rounding only)
    saved = torch.load(outs[1])
    for key, (x, k, bias) in saved["in"].items():
        ref = O.leaky_relu(O.conv2d_same(x.numpy(), k.numpy(), bias.numpy(), 1), 0.1)
        err = np.abs(saved["out"][key].numpy() - ref).max() / np.abs(ref).max()
        print(f"winograd kernel 4 vs oracle, {key}: max error / output range = {err:.2e}")
        assert err < 2e-5, (key, err)


def test_sncv_variants_are_bitwise_identical(dev, tmp_path):
    """m4d_sncv_fwd picks, by map size, the small-map kernel or the r = 3 tile kernel with its window rows split over
    1 / 2 / 4 lane groups; every choice must give the same bits.  The choices are read once per process, hence one
    subprocess per setting."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "sncv_variant_check.py")
    outs = []
    for name, env_add in (("ys1", {"M4D_SNCV_YS": "1", "M4D_SNCV_SMALL_PX": "0"}), ("ys2", {"M4D_SNCV_YS": "2", "M4D_SNCV_SMALL_PX": "0"}),
                          ("ys4", {"M4D_SNCV_YS": "4", "M4D_SNCV_SMALL_PX": "0"}), ("small", {"M4D_SNCV_SMALL_PX": "100000"}),
                          ("generic", {"M4D_SNCV_VARIANT": "0"})):
        out = str(tmp_path / f"{name}.pt")
        subprocess.run([sys.executable, tool, "--save", out], check=True, env=dict(os.environ, **env_add), timeout=600, capture_output=True)
        outs.append(out)
    res = subprocess.run([sys.executable, tool, "--compare", *outs], check=True, timeout=600, capture_output=True, text=True).stdout
    lines = [l for l in res.strip().splitlines() if l]
    assert len(lines) == 4 * 5 and all(l.endswith("bit-identical") for l in lines), res
    # ... and the tile kernel (ys1: small-map kernel disabled) against the oracle itself at these (C, k) geometries:
    # C = 96 / k = 4 and C = 192 / k = 8 otherwise only meet the oracle through the small-map kernel
    saved = torch.load(outs[0])
    for key, x in saved["in"].items():
        ref = O.cost_volume(x.numpy(), x.numpy(), 3, nbre_cuts=int(key.split("/")[1]))
        assert_bits_equal(saved["out"][key].numpy(), ref, f"SNCV tile kernel vs oracle, {key}")


@pytest.mark.parametrize("b,h,w,cin,cout,slope", [
    (1, 6, 20, 128, 128, 0.1),        # level-6 geometry
    (2, 12, 40, 240, 128, 0.1),       # level-5 first layer (padded refiner input), 15 chunks over 4 waves
    (1, 24, 80, 64, 32, 0.1),
    (1, 5, 7, 20, 40, 1.0),           # ragged tile, half-empty last chunk, Cout < CoutPad, 2 chunks (two waves idle)
    (3, 9, 11, 16, 5, 0.1),           # a single chunk
])
def test_small_map_conv(M, dev, b, h, w, cin, cout, slope):
    """m4d_conv3x3_small_bias_act (one launch, K split over the waves of a workgroup) vs the oracle; deterministic."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    wp, cpad = nops.pack_conv_weights(k)
    xd, wd, bd = to_dev(x, dev), to_dev(wp, dev), to_dev(bias, dev)
    got = nops.conv3x3_small_bias_act(xd, wd, bd, cout, cpad, slope)
    ref = O.conv2d_same(x, k, bias, 1)
    ref = np.where(ref > 0, ref, ref * F(slope)).astype(F)
    err = np.max(np.abs(npy(got) - ref))
    assert err < 1e-5 * max(1.0, np.abs(ref).max()), err
    assert torch.equal(got, nops.conv3x3_small_bias_act(xd, wd, bd, cout, cpad, slope))
    # the general entry (split-K + ordered reduce on such maps) agrees to rounding
    alt = nops.conv3x3_bias_act(xd, wd, bd, cout, cpad, slope)
    assert np.max(np.abs(npy(alt) - npy(got))) < 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("b,h,w,C,k", [(2, 12, 20, 128, 4), (1, 6, 20, 192, 8), (1, 90, 120, 32, 2)])
def test_merged_cost_volume_launch(M, dev, b, h, w, C, k):
    """m4d_dscv_sncv_fwd (one launch for both volumes on small maps, the two entries in sequence above 6000 pixels) gives the
    bits of m4d_dscv_fwd + m4d_sncv_fwd."""
    import ctypes
    from m4depth_amd._lib import lib, check, dptr, stream_ptr
    rng = np.random.default_rng(C + k)
    cam = camera_np(b, h, w)
    rot, trans = motion_np(rng, b, t_scale=(3.0, 3.0, 1.0))
    c1 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), k)
    c2 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), k)
    disp = (0.2 + 6.0 * rng.random([b, h, w, 1])).astype(F)
    dpt = (0.2 + 6.0 * rng.random([b, h, w, 1])).astype(F)
    t = {n: to_dev(v, dev) for n, v in dict(c1=c1, c2=c2, disp=disp, dpt=dpt, rot=rot, trans=trans, f=cam["f"], c=cam["c"]).items()}
    stride = 9 * k + 49 * k + 3
    outs = []
    for merged in (True, False):
        buf = torch.zeros(b, h, w, stride, device=dev)
        base = buf.data_ptr()
        args = (dptr(t["c1"]), dptr(t["c2"]), dptr(t["dpt"]), dptr(t["disp"]), dptr(t["rot"]), t["rot"].shape[1], dptr(t["trans"]),
                dptr(t["f"]), dptr(t["c"]), b, h, w, C, 4, k, 0, ctypes.c_void_p(base), stride, None, None, 1, 1.0)
        if merged:
            check(lib.m4d_dscv_sncv_fwd(*args, 3, ctypes.c_void_p(base + 4 * (9 * k + 1)), stride, stream_ptr()), "m4d_dscv_sncv_fwd")
        else:
            check(lib.m4d_dscv_fwd(*args, None, stream_ptr()), "m4d_dscv_fwd")
            check(lib.m4d_sncv_fwd(dptr(t["c1"]), dptr(t["c1"]), b, h, w, C, 3, 1, k, ctypes.c_void_p(base + 4 * (9 * k + 1)), stride,
                                   stream_ptr()), "m4d_sncv_fwd")
        outs.append(buf)
    assert torch.equal(outs[0], outs[1])
    ocv, _ = O.get_parallax_sweeping_cv(c1, c2, dpt, disp, rot, trans, cam, 4, k)
    assert_bits_equal(npy(outs[0][..., :9 * k]), ocv, "dscv through the merged entry")
    assert_bits_equal(npy(outs[0][..., 9 * k + 1:9 * k + 1 + 49 * k]), O.cost_volume(c1, c1, 3, nbre_cuts=k), "sncv through the merged entry")


@pytest.mark.parametrize("depth,b,h,w,quat,cv_accum", [
    (1, 2, 37, 45, True, "fp32_round"),      # C = 16, 1 cut: ragged tiles in both directions
    (1, 1, 8, 32, False, "fp16_seq"),        # exactly one tile; small-angle rotation; sequential float16 mean
    (2, 2, 20, 27, True, "fp32_round"),      # C = 32, 2 cuts, padded row stride 122 -> 128
    (3, 1, 19, 33, True, "fp32_round"),      # C = 64, 2 cuts
    (3, 2, 16, 16, False, "fp16_seq"),
    (4, 2, 21, 19, True, "fp32_round"),      # C = 96, 4 cuts of 24 (24 lanes per pixel, 6 per cut): ragged 8x8 tiles, large-map shape
    (4, 1, 8, 12, False, "fp16_seq"),        #   ... the half-size tiles of small maps
    (5, 2, 12, 17, True, "fp32_round"),      # C = 128, 4 cuts of 32
    (5, 1, 9, 40, True, "fp16_seq"),
    (6, 2, 6, 20, True, "fp32_round"),       # C = 192, 8 cuts of 24 (48 lanes per pixel): the coarsest level, no coarser estimate at step 3
    (6, 3, 13, 11, False, "fp32_round"),
])
def test_fused_level_front_is_bitwise_the_separate_kernels(M, dev, depth, b, h, w, quat, cv_accum):
    """m4d_level_front (normalise + level_pre + DSCV + SNCV in one launch, whole refiner-input rows) against the three
    separate launches it replaces: refiner input (incl. the zero padding channels), feature state and outputs bit for bit,
    over three consecutive frames (the second and third see a depth memory and a coarser estimate that vary per pixel)."""
    _fused_front_vs_separate(M, dev, depth, b, h, w, quat, cv_accum, 4, 3)


@pytest.mark.parametrize("depth,b,h,w,quat,cv_accum", [
    (1, 1, 21, 37, True, "fp32_round"),      # 16x8 tiles, ragged both ways; 13 hypotheses over 4 lanes per pixel
    (1, 2, 8, 16, False, "fp16_seq"),        # exactly one tile
    (2, 1, 19, 20, True, "fp32_round"),      # C = 32, 2 cuts: 8x8 tiles, refiner-input rows of 370 (stride 376)
    (3, 2, 16, 17, True, "fp32_round"),      # C = 64: the 109-KB halo
    (3, 1, 9, 11, False, "fp16_seq"),
    (4, 1, 13, 18, True, "fp32_round"),      # C = 96, 4 cuts of 24: 8x4 tiles, rows of 734
    (5, 2, 7, 10, True, "fp32_round"),       # C = 128: 4x4 tiles
    (5, 1, 8, 8, False, "fp16_seq"),
])
def test_fused_level_front_large_windows_is_bitwise_the_separate_kernels(M, dev, depth, b, h, w, quat, cv_accum):
    """BASELINE configs[4] (search ranges 6 / 6: 13 DSCV hypotheses, a 13x13 SNCV window -- "large-window LDS tiling stress"):
    m4d_level_front_r's (6, 6) instantiations (levels 1-5, round 5) against m4d_level_pre_normalize + m4d_dscv_fwd +
    m4d_sncv_fwd with the same ranges, bit for bit; level 6 (192 channels: a 200-KB halo) has no fused form and says so."""
    from m4depth_amd import network as net
    assert not net.lib.m4d_level_front_supported(192, 8, 6, 6, (182 * 8 + 6 + 7) // 8 * 8)
    _fused_front_vs_separate(M, dev, depth, b, h, w, quat, cv_accum, 6, 6)


@pytest.mark.parametrize("depth,b,h,w,quat,cv_accum", [
    (4, 1, 24, 80, True, "fp32_round"),      # levels 4-6 of the 384x1280 pyramid at batch 1: what the bench path launches
    (5, 1, 12, 40, True, "fp32_round"),
    (6, 1, 6, 20, True, "fp32_round"),
    (4, 2, 21, 19, False, "fp16_seq"),       # ragged sizes, small-angle rotation, sequential float16 mean, batch 2
    (6, 3, 13, 11, True, "fp16_seq"),
    (3, 2, 9, 14, True, "fp32_round"),       # C = 64 / 2 cuts, (2, 2): the 3-level BASELINE configs[0] pyramid's coarse shapes
    (2, 1, 17, 23, False, "fp32_round"),     # C = 32 / 2 cuts
    (1, 1, 10, 12, True, "fp32_round"),      # C = 16 / 1 cut
])
def test_level_front_small_is_bitwise_the_separate_launches(M, dev, depth, b, h, w, quat, cv_accum, monkeypatch):
    """m4d_level_front_small (round 6, VERDICT r5 item 6: a coarse level opens with ONE launch -- level_pre's glue + DSCV + SNCV on
    features normalised ahead of the level by m4d_normalize_levels) against m4d_level_pre_normalize + m4d_dscv_sncv_fwd, the
    two dependent launches it replaces: refiner input (padding channels included), feature state, depth state and outputs bit
    for bit over four frames (reset, two full frames with a per-pixel depth memory and coarser estimate, one without a coarser
    level = the coarsest level's form of the call)."""
    from m4depth_amd import network as net, network_ops as nops, synthetic as S
    rng = np.random.default_rng(900 + depth * 10 + h)
    C = S_ENC[depth - 1]
    W = S.init_weights(6, seed=4)
    settings = {"nbre_lvls": 6, "is_training": False, "ablation": M.M4depthAblationParameters(), "cv_accum": cv_accum,
                "dscv_range": 4, "sncv_range": 3}
    monkeypatch.setattr(net, "fused_front_min_pixels", 10 ** 9)          # no big-tile fused front at any level here
    levels = []
    for _ in range(2):
        gl = M.DepthEstimatorLevel(settings, depth)
        convs = list(gl.disp_refiner.prep_conv_layers) + list(gl.disp_refiner.est_d_conv_layers)
        for i, cv in enumerate(convs):
            cv.load_hwio(W[f"lvl.{depth}.conv.{i}.kernel"], W[f"lvl.{depth}.conv.{i}.bias"], dev)
        levels.append(gl)
    assert levels[0].wants_prenormalized(b, h, w, C)
    cam = to_dev(camera_np(b, h, w), dev)
    launches = []
    for step in range(4):
        rot, trans = motion_np(rng, b, quat=quat, t_scale=(3.0, 3.0, 1.0))
        f = to_dev(rng.standard_normal([b, h, w, C]).astype(F), dev)
        prev = {"depth": to_dev((1 + 50 * rng.random([b, (h + 1) // 2, (w + 1) // 2, 1])).astype(F), dev),
                "parallax": to_dev((0.2 + 2 * rng.random([b, (h + 1) // 2, (w + 1) // 2, 1])).astype(F), dev),
                "other": to_dev(rng.standard_normal([b, (h + 1) // 2, (w + 1) // 2, 4]).astype(F), dev)}
        if step == 3:
            prev = None
        nt = np.full([b], step == 0)
        nf = nops.normalize_levels([(f, levels[0].nbre_cuts)])[0]
        assert torch.equal(nf, nops.normalize_cuts(f, levels[0].nbre_cuts)), "m4d_normalize_levels vs m4d_normalize_cuts"
        outs, fins = [], []
        for small, gl in zip((True, False), levels):
            n0 = int(net.lib.m4d_launch_count())
            outs.append(gl(f, prev, to_dev(rot, dev), to_dev(trans, dev), cam, nt, curr_f_normalized=nf if small else None))
            launches.append(int(net.lib.m4d_launch_count()) - n0)
            fins.append(gl.last_f_input.clone() if step > 0 else None)
        assert torch.equal(levels[0].prev_f_maps, levels[1].prev_f_maps), f"step {step}: feature state"
        assert torch.equal(levels[0].depth_prev_t, levels[1].depth_prev_t), f"step {step}: depth state"
        for key in ("depth", "parallax", "other"):
            assert torch.equal(outs[0][key], outs[1][key]), f"step {step}: {key}"
        if step > 0:
            assert launches[-2] == launches[-1] - 1, f"step {step}: {launches[-2]} launches against {launches[-1]}"
            assert_bits_equal(npy(fins[0]), npy(fins[1]), f"step {step}: refiner input")
            assert torch.isfinite(fins[0]).all()


def _fused_front_vs_separate(M, dev, depth, b, h, w, quat, cv_accum, rd, rs):
    from m4depth_amd import network as net
    rng = np.random.default_rng(500 + depth * 10 + h)
    C = S_ENC[depth - 1]
    from m4depth_amd import synthetic as S
    W = S.init_weights(6, seed=4, dscv_range=rd, sncv_range=rs)
    settings = {"nbre_lvls": 6, "is_training": False, "ablation": M.M4depthAblationParameters(), "cv_accum": cv_accum,
                "dscv_range": rd, "sncv_range": rs}
    levels = []
    for _ in range(2):
        gl = M.DepthEstimatorLevel(settings, depth)
        convs = list(gl.disp_refiner.prep_conv_layers) + list(gl.disp_refiner.est_d_conv_layers)
        for i, cv in enumerate(convs):
            cv.load_hwio(W[f"lvl.{depth}.conv.{i}.kernel"], W[f"lvl.{depth}.conv.{i}.bias"], dev)
        levels.append(gl)
    cam = to_dev(camera_np(b, h, w), dev)
    old = (net.fused_front, net.fused_front_min_pixels, net.fused_front_coarse_min_pixels)
    net.fused_front_coarse_min_pixels = 0            # levels 4-6: take the fused front on these small maps too
    f_row = (2 * rd + 1 + (2 * rs + 1) ** 2) * 2 ** (depth // 2) + 6
    assert bool(net.lib.m4d_level_front_supported(C, 2 ** (depth // 2), rd, rs, (f_row + 7) // 8 * 8))
    try:
        for step in range(4):
            rot, trans = motion_np(rng, b, quat=quat, t_scale=(3.0, 3.0, 1.0))
            f = to_dev(rng.standard_normal([b, h, w, C]).astype(F), dev)
            prev = {"depth": to_dev((1 + 50 * rng.random([b, (h + 1) // 2, (w + 1) // 2, 1])).astype(F), dev),
                    "parallax": to_dev((0.2 + 2 * rng.random([b, (h + 1) // 2, (w + 1) // 2, 1])).astype(F), dev),
                    "other": to_dev(rng.standard_normal([b, (h + 1) // 2, (w + 1) // 2, 4]).astype(F), dev)}
            if step == 3:
                prev = None                      # the coarsest-level form of the call (no coarser estimate)
            nt = np.full([b], step == 0)
            outs, fins = [], []
            for fused, gl in zip((True, False), levels):
                net.fused_front, net.fused_front_min_pixels = fused, 0
                outs.append(gl(f, prev, to_dev(rot, dev), to_dev(trans, dev), cam, nt))
                fins.append(gl.last_f_input.clone() if step > 0 else None)     # both levels share the persistent padded buffer
            assert torch.equal(levels[0].prev_f_maps, levels[1].prev_f_maps), f"step {step}: feature state"
            assert torch.equal(levels[0].depth_prev_t, levels[1].depth_prev_t), f"step {step}: depth state"
            for key in ("depth", "parallax", "other"):
                assert torch.equal(outs[0][key], outs[1][key]), f"step {step}: {key}"
            if step > 0:
                fa, fb = fins
                assert_bits_equal(npy(fa), npy(fb), f"step {step}: refiner input")
                assert torch.isfinite(fa).all()
    finally:
        net.fused_front, net.fused_front_min_pixels, net.fused_front_coarse_min_pixels = old


S_ENC = [16, 32, 64, 96, 128, 192]


@pytest.mark.parametrize("b,h,w,cin,cout", [(1, 64, 96, 128, 96), (1, 64, 96, 64, 128), (2, 33, 47, 48, 128), (1, 50, 70, 32, 100),
                                             (3, 16, 16, 16, 96), (1, 17, 31, 96, 68), (1, 20, 37, 32, 64), (2, 9, 17, 16, 38),
                                             (1, 48, 160, 128, 128)])
def test_wide_bf16_split_winograd_is_bit_identical(dev, b, h, w, cin, cout):
    """(Also the half-tile kernel m4d_wino6h.hip, 16x8 pixels x 64 output channels per workgroup, one pass, wave-private
    fragment rings and three raw-halo buffers: the same bits again.)  m4d_wino6w.hip (16x16 pixels x all 96 / 128 output channels per workgroup, two passes over the Winograd position
    rows, the first pass's partial row transform parked in the output pixels) against m4d_wino6.hip (x 64 output channels,
    one pass): the same float32 bits -- same products, same accumulation order, same association in the output transform --
    on whole / ragged / odd-sized maps, batches, 3 and 4 N-tiles, channel counts that are not multiples of 32; and the
    same bits again through the default dispatch."""
    from m4depth_amd import network_ops as nops
    from m4depth_amd import _lib
    from m4depth_amd._lib import lib
    if not _lib.has_experiments:
        pytest.skip("m4d_wino6w.hip / m4d_wino6h.hip are experiments: make EXPERIMENTS=1 (include/m4depth_hip_experiments.h)")
    rng = np.random.default_rng(b * 1000 + h + cin + cout)
    x = to_dev(rng.standard_normal([b, h, w, cin]).astype(F), dev)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = to_dev((0.1 * rng.standard_normal([cout])).astype(F), dev)
    wu6, cpad = nops.pack_conv_weights_wino6(k)
    wud = torch.from_numpy(wu6.view("int16")).to(dev)
    try:
        lib.m4d_wino6_set_variant(1)
        narrow = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1)
        lib.m4d_wino6_set_variant(2)
        wide = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1)
        wide2 = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1)
        lib.m4d_wino6_set_variant(3)                # the half-tile kernel (m4d_wino6h.hip): 16x8 pixels x 64 couts per workgroup
        half = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1)
        lib.m4d_wino6_set_variant(0)
        lib.m4d_wino6_set_half_tile_max_workgroups(1 << 20)       # ... and through the grid-size rule of the default variant
        half2 = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1)
    finally:
        lib.m4d_wino6_set_variant(0)
        lib.m4d_wino6_set_half_tile_max_workgroups(0)
    auto = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1)
    assert torch.equal(wide, narrow), f"{int((wide != narrow).sum())} of {narrow.numel()} elements differ"
    assert torch.equal(half, narrow), f"half-tile kernel: {int((half != narrow).sum())} of {narrow.numel()} elements differ"
    assert torch.equal(wide2, wide) and torch.equal(auto, narrow) and torch.equal(half2, half)
    ref = O.leaky_relu(O.conv2d_same(npy(x), k, npy(bias), 1), 0.1)
    assert np.max(np.abs(npy(wide) - ref)) < 1e-5 * max(1.0, np.abs(ref).max())


def test_wide_bf16_split_winograd_level1_layer(dev):
    """The 96-wide layer of level 1 (192x640: 480 tiles, 3 N-tiles) through the wide kernel (variant 2): bits equal to the
    default kernel's."""
    from m4depth_amd import network_ops as nops
    from m4depth_amd import _lib
    from m4depth_amd._lib import lib
    if not _lib.has_experiments:
        pytest.skip("m4d_wino6w.hip is an experiment: make EXPERIMENTS=1 (include/m4depth_hip_experiments.h)")
    rng = np.random.default_rng(5)
    x = to_dev(rng.standard_normal([1, 192, 640, 128]).astype(F), dev)
    k = (rng.standard_normal([3, 3, 128, 96]) * np.sqrt(2.0 / (9 * 128))).astype(F)
    bias = to_dev((0.1 * rng.standard_normal([96])).astype(F), dev)
    wu6, cpad = nops.pack_conv_weights_wino6(k)
    wud = torch.from_numpy(wu6.view("int16")).to(dev)
    auto = nops.conv3x3_wino6_bias_act(x, wud, bias, 96, cpad, 0.1)
    try:
        lib.m4d_wino6_set_variant(2)
        wide = nops.conv3x3_wino6_bias_act(x, wud, bias, 96, cpad, 0.1)
    finally:
        lib.m4d_wino6_set_variant(0)
    assert torch.equal(auto, wide)
