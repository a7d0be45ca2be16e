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
