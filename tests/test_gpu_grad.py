"""Backward kernels of the cost volumes (m4d_dscv_bwd / m4d_sncv_bwd through the
reference-named differentiable ops) vs autodiff of the torch-CPU restatement of the same
TF graph (oracle/m4depth_oracle_train.py).

Tolerances: the SNCV gradient is float32 sums of <= 49 products: 1e-5 relative to the
largest gradient.  The DSCV gradient passes through float16 multiplies (relative step 1e-3)
and an atomic scatter: 4e-3 of the largest gradient per tensor.
"""
import numpy as np
import pytest
import torch

from oracle import m4depth_oracle as O
from oracle import m4depth_oracle_train as OT
from helpers import F, camera_np, motion_np, to_dev, npy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import m4depth_amd
    return m4depth_amd


def _close(got, want, tol, what):
    scale = float(np.max(np.abs(want))) + 1e-30
    err = float(np.max(np.abs(got - want))) / scale
    assert err < tol, f"{what}: max |diff| / max |grad| = {err:.3e} (tol {tol})"


@pytest.mark.parametrize("C,k,r,h,w", [(16, 1, 4, 24, 40), (32, 2, 4, 12, 20), (96, 4, 2, 6, 10), (192, 8, 4, 6, 10)])
def test_dscv_backward(M, dev, C, k, r, h, w):
    rng = np.random.default_rng(C + k)
    b = 2
    c1 = O.normalize_cuts(rng.normal(size=[b, h, w, C]).astype(F), k)
    c2 = O.normalize_cuts(rng.normal(size=[b, h, w, C]).astype(F), k)
    disp = rng.uniform(0.5, 6.0, [b, h, w, 1]).astype(F)
    dpt = rng.uniform(0.5, 6.0, [b, h, w, 1]).astype(F)
    rot, trans = motion_np(rng, b)
    cam = camera_np(b, h, w)
    ncp = 2 * r + 1
    g_cv = rng.normal(size=[b, h, w, k * ncp]).astype(F)
    g_pd = rng.normal(size=[b, h, w, ncp]).astype(F)

    tc = [torch.from_numpy(a).requires_grad_(True) for a in (c1, c2, dpt, disp)]
    cv, pd = OT.get_parallax_sweeping_cv(*tc, rot, trans, cam, r, nbre_cuts=k)
    (cv * torch.from_numpy(g_cv)).sum().add((pd * torch.from_numpy(g_pd)).sum()).backward()

    dc = [to_dev(a, dev).requires_grad_(True) for a in (c1, c2, dpt, disp)]
    cv_d, pd_d = M.get_parallax_sweeping_cv(*dc, to_dev(rot, dev), to_dev(trans, dev), to_dev(cam, dev), r, nbre_cuts=k)
    ((cv_d * to_dev(g_cv, dev)).sum() + (pd_d * to_dev(g_pd, dev)).sum()).backward()
    for name, t_cpu, t_dev in zip(("g_c1", "g_c2", "g_disp_prev_t", "g_disp"), tc, dc):
        assert t_dev.grad is not None, name
        _close(npy(t_dev.grad), t_cpu.grad.numpy(), 4e-3, f"dscv {name} C={C} k={k}")


@pytest.mark.parametrize("C,k,r,same", [(16, 1, 3, True), (32, 2, 3, True), (32, 2, 2, False), (96, 4, 3, True)])
def test_sncv_backward(M, dev, C, k, r, same):
    rng = np.random.default_rng(C + r)
    b, h, w = 2, 10, 14
    c1 = O.normalize_cuts(rng.normal(size=[b, h, w, C]).astype(F), k)
    c2 = c1 if same else O.normalize_cuts(rng.normal(size=[b, h, w, C]).astype(F), k)
    g = rng.normal(size=[b, h, w, (2 * r + 1) ** 2 * k]).astype(F)

    t1 = torch.from_numpy(c1).requires_grad_(True)
    t2 = t1 if same else torch.from_numpy(c2).requires_grad_(True)
    (OT.cost_volume(t1, t2, r, nbre_cuts=k) * torch.from_numpy(g)).sum().backward()
    d1 = to_dev(c1, dev).requires_grad_(True)
    d2 = d1 if same else to_dev(c2, dev).requires_grad_(True)
    (M.cost_volume(d1, d2, r, nbre_cuts=k) * to_dev(g, dev)).sum().backward()
    _close(npy(d1.grad), t1.grad.numpy(), 1e-5, f"sncv g_c1 C={C} k={k} same={same}")
    if not same:
        _close(npy(d2.grad), t2.grad.numpy(), 1e-5, f"sncv g_c2 C={C} k={k}")


def test_dscv_backward_no_grad_paths(M, dev):
    """disp_prev_t comes out of prev_d2para's stop_gradient in the model (depth_operations.py:215):
    the op must work when only some inputs record gradients, and stay on the plain path otherwise."""
    rng = np.random.default_rng(2)
    b, h, w, C, k, r = 1, 8, 12, 16, 1, 4
    c1 = to_dev(O.normalize_cuts(rng.normal(size=[b, h, w, C]).astype(F), k), dev)
    c2 = to_dev(O.normalize_cuts(rng.normal(size=[b, h, w, C]).astype(F), k), dev)
    disp = to_dev(rng.uniform(0.5, 6.0, [b, h, w, 1]).astype(F), dev).requires_grad_(True)
    dpt = to_dev(rng.uniform(0.5, 6.0, [b, h, w, 1]).astype(F), dev)
    rot, trans = motion_np(rng, b)
    cam = to_dev(camera_np(b, h, w), dev)
    cv, pd = M.get_parallax_sweeping_cv(c1, c2, dpt, disp, to_dev(rot, dev), to_dev(trans, dev), cam, r)
    cv.sum().backward()
    assert disp.grad is not None and torch.isfinite(disp.grad).all()
    with torch.no_grad():
        cv2, _ = M.get_parallax_sweeping_cv(c1, c2, dpt, disp, to_dev(rot, dev), to_dev(trans, dev), cam, r)
    assert not cv2.requires_grad and torch.equal(cv2, cv.detach())


# ------------------------------------------------------------------ training glue kernels
def test_pack_conv_weights_kernel_matches_host_packer(M, dev):
    from m4depth_amd import network_ops as nops, training as TR
    rng = np.random.default_rng(0)
    for O, I in ((45, 37), (128, 122), (5, 16)):
        w = torch.from_numpy(rng.normal(size=[O, I, 3, 3]).astype(F)).to(dev).contiguous(memory_format=torch.channels_last)
        hwio = npy(w).transpose(2, 3, 1, 0)
        want, cpad = nops.pack_conv_weights(hwio)
        got, npad = TR._packed({}, w, False)
        assert npad == cpad and np.array_equal(npy(got), want)
        want_t, cpad_t = nops.pack_conv_weights(np.ascontiguousarray(hwio[::-1, ::-1].transpose(0, 1, 3, 2)))
        got_t, npad_t = TR._packed({}, w, True)
        assert npad_t == cpad_t and np.array_equal(npy(got_t), want_t)


def test_pack_conv_weights_lat_kernel_matches_host_packer(M, dev):
    """m4d_pack_conv_weights_lat (the per-step device packer of the training path: fragment-major layout of m4d_conv3x3_lat,
    every weight split exactly into three bf16 terms) against network_ops.pack_conv_weights_lat, forward and data-gradient
    (rotated / transposed) forms, bit for bit."""
    from m4depth_amd import network_ops as nops, training as TR
    rng = np.random.default_rng(1)
    for O, I in ((45, 37), (128, 122), (5, 16), (96, 128)):
        w = torch.from_numpy(rng.normal(size=[O, I, 3, 3]).astype(F)).to(dev).contiguous(memory_format=torch.channels_last)
        hwio = npy(w).transpose(2, 3, 1, 0)
        want = nops.pack_conv_weights_lat(hwio).view(np.int16)
        got = TR._packed_lat({}, w, False)
        assert got.shape == want.shape and np.array_equal(npy(got), want)
        want_t = nops.pack_conv_weights_lat(np.ascontiguousarray(hwio[::-1, ::-1].transpose(0, 1, 3, 2))).view(np.int16)
        got_t = TR._packed_lat({}, w, True)
        assert got_t.shape == want_t.shape and np.array_equal(npy(got_t), want_t)


def test_glue_backward_kernels_match_autodiff_of_the_restatement(M, dev):
    from m4depth_amd import training as TR
    rng = np.random.default_rng(4)
    b, h, w = 2, 6, 9
    # x2 legacy upsample
    x = rng.normal(size=[b, h, w, 4]).astype(F)
    g = rng.normal(size=[b, 2 * h, 2 * w, 4]).astype(F)
    tx = torch.from_numpy(x).requires_grad_(True)
    (OT.resize_bilinear(tx, 2 * h, 2 * w, False) * 2.0 * torch.from_numpy(g)).sum().backward()
    dx = to_dev(x, dev).requires_grad_(True)
    (TR._upsample2_v1(dx, 2 * h, 2 * w, 2.0) * to_dev(g, dev)).sum().backward()
    _close(npy(dx.grad), tx.grad.numpy(), 1e-6, "upsample adjoint")
    # per-cut normalisation
    x = rng.normal(size=[b, h, w, 32]).astype(F)
    g = rng.normal(size=[b, h, w, 32]).astype(F)
    tx = torch.from_numpy(x).requires_grad_(True)
    (OT.normalize_cuts(tx, 2) * torch.from_numpy(g)).sum().backward()
    dx = to_dev(x, dev).requires_grad_(True)
    (TR._normalize_cuts(dx, 2) * to_dev(g, dev)).sum().backward()
    _close(npy(dx.grad), tx.grad.numpy(), 1e-5, "normalize_cuts backward")
    # level tail
    ro = rng.normal(0, 3.0, size=[b, h, w, 5]).astype(F)
    ro[0, 0, 0, 0] = 9.0                                   # outside the clip: no gradient
    rot, trans = motion_np(rng, b)
    cam = camera_np(b, h, w)
    gs = [rng.normal(size=[b, h, w, n]).astype(F) for n in (1, 1, 4)]
    tro = torch.from_numpy(ro).requires_grad_(True)
    para = torch.exp(torch.clamp(tro[..., :1], -7., 7.)) / 0.5
    depth = OT.parallax2depth(para, rot, trans, cam)
    ((para * torch.from_numpy(gs[0])).sum() + (depth * torch.from_numpy(gs[1])).sum()
     + (tro[..., 1:] * torch.from_numpy(gs[2])).sum()).backward()
    dro = to_dev(ro, dev).requires_grad_(True)
    dcam = to_dev(cam, dev)
    p_d, d_d, o_d = TR._LevelPost.apply(dro, to_dev(rot, dev), to_dev(trans, dev), dcam["f"], dcam["c"], 0.5)
    ((p_d * to_dev(gs[0], dev)).sum() + (d_d * to_dev(gs[1], dev)).sum() + (o_d * to_dev(gs[2], dev)).sum()).backward()
    _close(npy(dro.grad), tro.grad.numpy(), 1e-5, "level tail backward")
    assert npy(dro.grad)[0, 0, 0, 0] == 0.0
    # loss term of one level, both ground-truth kinds
    gt = rng.uniform(1, 80, size=[b, 4 * h, 4 * w, 1]).astype(F)
    pred = rng.uniform(0.005, 250, size=[b, h, w, 1]).astype(F)
    for kind in ("map", "velodyne"):
        gtk = gt * (rng.random(gt.shape) > 0.8) if kind == "velodyne" else gt
        gtk = gtk.astype(F)
        tp = torch.from_numpy(pred).requires_grad_(True)
        ref = OT.m4depth_loss([None, {"depth": torch.from_numpy(gtk)}], [None, [{"depth": tp}]], kind)
        ref.backward()
        dp = to_dev(pred, dev).requires_grad_(True)
        got = TR.m4depth_loss([None, {"depth": to_dev(gtk, dev)}], [None, [{"depth": dp}]], kind)
        got.backward()
        assert abs(got.item() - ref.item()) <= 2e-6 * abs(ref.item()), (kind, got.item(), ref.item())
        _close(npy(dp.grad), tp.grad.numpy(), 1e-5, f"loss backward ({kind})")


def test_bias_act_backward_kernel(M, dev):
    from m4depth_amd._lib import lib, dptr, stream_ptr, check
    rng = np.random.default_rng(8)
    for rows, C in ((1000, 5), (777, 96), (300, 128), (64, 470)):
        g = rng.normal(size=[rows, C]).astype(F)
        out = rng.normal(size=[rows, C]).astype(F)
        dg, dout = to_dev(g, dev), to_dev(out, dev)
        gp = torch.empty_like(dg)
        gb = torch.empty(C, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.m4d_bias_act_bwd_workspace_floats(rows, C)), dtype=torch.float32, device=dev)
        check(lib.m4d_bias_act_bwd(dptr(dg), dptr(dout), rows, C, 0.1, dptr(gp), dptr(gb), dptr(ws), stream_ptr()), "bwd")
        want = g * np.where(out > 0, F(1.0), F(0.1))
        assert np.array_equal(npy(gp), want)
        np.testing.assert_allclose(npy(gb), want.sum(axis=0, dtype=np.float64), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("b,h,w,cin,cout,stride", [
    (2, 16, 24, 32, 64, 1),       # whole 32-blocks (a small map: forward and data gradient on the latency-first kernel)
    (3, 24, 24, 128, 96, 1),      # level 4 of a 384x384 training crop at batch 3: the same kernels, partly filled last cout group
    (1, 40, 60, 32, 64, 1),       # above the small-map threshold: the fp32-MFMA pair
    (1, 9, 21, 122, 96, 1),       # ragged tiles, Cin not a multiple of 4 (the training graph's exact-width refiner input)
    (3, 12, 16, 16, 16, 2),       # stride 2, even size: TF pads bottom / right only
    (1, 11, 13, 64, 64, 2),       # stride 2, odd sizes: one before, one after
    (2, 20, 28, 3, 16, 1),        # the 3-channel image layer
    (1, 6, 10, 470, 128, 1),      # the widest refiner input
])
def test_conv_backward_kernels_match_float64_autodiff(M, dev, b, h, w, cin, cout, stride):
    """training._ConvBiasAct (forward MFMA convolution; backward = bias/activation kernel, data gradient through the MFMA
    convolution on rotated weights (+ m4d_dilate2 for stride 2), weight gradient through m4d_conv3x3_wgrad) against
    float64 autodiff of the same TF-'SAME' convolution on the CPU.  Tolerance: 2e-5 of the largest gradient entry (float32
    sums over up to b*h*w pixels / 9*Cout products)."""
    import torch.nn.functional as TF
    from m4depth_amd import training as TR
    rng = np.random.default_rng(cin * 3 + cout + stride)
    x = rng.standard_normal([b, h, w, cin]).astype(F)
    k = (rng.standard_normal([cout, cin, 3, 3]) * np.sqrt(2.0 / (9 * cin))).astype(F)
    bias = (0.1 * rng.standard_normal([cout])).astype(F)
    oh, ow = -(-h // stride), -(-w // stride)
    gout = rng.standard_normal([b, oh, ow, cout]).astype(F)
    # float64 reference
    xr = torch.from_numpy(x).double().requires_grad_(True)
    kr = torch.from_numpy(k).double().requires_grad_(True)
    br = torch.from_numpy(bias).double().requires_grad_(True)
    ph = max((oh - 1) * stride + 3 - h, 0)
    pw = max((ow - 1) * stride + 3 - w, 0)
    xp = TF.pad(xr.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    yr = TF.leaky_relu(TF.conv2d(xp, kr, br, stride, 0), 0.1).permute(0, 2, 3, 1)
    (yr * torch.from_numpy(gout).double()).sum().backward()
    # the product's autograd node
    xd = to_dev(x, dev).requires_grad_(True)
    wd = torch.nn.Parameter(to_dev(k, dev).contiguous(memory_format=torch.channels_last))
    bd = torch.nn.Parameter(to_dev(bias, dev))
    yd = TR._ConvBiasAct.apply(xd, wd, bd, stride, 0.1, {})
    np.testing.assert_allclose(npy(yd), yr.detach().numpy(), atol=2e-5 * float(yr.abs().max()))
    (yd * to_dev(gout, dev)).sum().backward()
    assert wd.grad.shape == wd.shape
    _close(npy(wd.grad), kr.grad.numpy(), 2e-5, "weight gradient")
    _close(npy(bd.grad), br.grad.numpy(), 2e-5, "bias gradient")
    _close(npy(xd.grad), xr.grad.numpy(), 2e-5, "data gradient")
    # deterministic: a second backward gives the same bits
    wd.grad = None
    xd.grad = None
    yd2 = TR._ConvBiasAct.apply(xd, wd, bd, stride, 0.1, {})
    (yd2 * to_dev(gout, dev)).sum().backward()
    assert torch.equal(yd2, yd)
    g1 = wd.grad.clone()
    wd.grad = None
    yd3 = TR._ConvBiasAct.apply(xd, wd, bd, stride, 0.1, {})
    (yd3 * to_dev(gout, dev)).sum().backward()
    assert torch.equal(wd.grad, g1)
