"""The differentiable torch-CPU restatement (oracle/m4depth_oracle_train.py) against the
strict-float32 numpy oracle: same forward values, so that its autodiff gradients are the
gradients of the graph the parity tests pin."""
import numpy as np
import pytest
import torch

from oracle import m4depth_oracle as O
from oracle import m4depth_oracle_train as OT
from m4depth_amd import synthetic
from helpers import camera_np, motion_np

F = np.float32


def _unit(rng, shape, k):
    x = rng.normal(size=shape).astype(F)
    return O.normalize_cuts(x, k)


def test_dscv_and_sncv_forward_match_numpy_oracle():
    rng = np.random.default_rng(5)
    b, h, w, C, k = 2, 12, 20, 16, 2
    c1, c2 = _unit(rng, [b, h, w, C], k), _unit(rng, [b, h, w, C], k)
    disp = rng.uniform(0.5, 6.0, [b, h, w, 1]).astype(F)
    dpt = rng.uniform(0.5, 6.0, [b, h, w, 1]).astype(F)
    rot, trans = motion_np(rng, b)
    cam = camera_np(b, h, w)
    cv, pd = O.get_parallax_sweeping_cv(c1, c2, dpt, disp, rot, trans, cam, 4, nbre_cuts=k)
    tcv, tpd = OT.get_parallax_sweeping_cv(*(torch.from_numpy(a) for a in (c1, c2, dpt, disp)), rot, trans, cam, 4,
                                           nbre_cuts=k)
    # the float16 mean: torch accumulates in float and rounds once (= cv_accum "fp32_round"), possibly in
    # another order than the sequential oracle -> one half ulp
    np.testing.assert_allclose(tcv.numpy(), cv, rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(tpd.numpy(), pd, rtol=1e-5, atol=1e-6)
    sn = O.cost_volume(c1, c1, 3, nbre_cuts=k)
    tsn = OT.cost_volume(torch.from_numpy(c1), torch.from_numpy(c1), 3, nbre_cuts=k)
    np.testing.assert_allclose(tsn.numpy(), sn, rtol=1e-5, atol=1e-6)


def test_resizes_and_converters_match_numpy_oracle():
    rng = np.random.default_rng(6)
    x = rng.normal(size=[2, 5, 7, 3]).astype(F)
    np.testing.assert_allclose(OT.resize_bilinear(torch.from_numpy(x), 10, 14, False).numpy(),
                               O.resize_bilinear_v1(x, 10, 14), rtol=1e-6, atol=1e-6)
    # half-pixel downscale by 2 = 2x2 box mean (tf.image.resize bilinear, m4depth_network.py:532)
    y = rng.normal(size=[1, 8, 12, 1]).astype(F)
    want = y.reshape(1, 4, 2, 6, 2, 1).mean(axis=(2, 4))
    np.testing.assert_allclose(OT.resize_bilinear(torch.from_numpy(y), 4, 6, True).numpy(), want, rtol=1e-6, atol=1e-6)
    b, h, w = 2, 6, 9
    rot, trans = motion_np(rng, b)
    cam = camera_np(b, h, w)
    d = rng.uniform(1, 80, [b, h, w, 1]).astype(F)
    np.testing.assert_allclose(OT.depth2parallax(torch.from_numpy(d), rot, trans, cam).numpy(),
                               O.depth2parallax(d, rot, trans, cam), rtol=1e-6)
    np.testing.assert_allclose(OT.prev_d2para(torch.from_numpy(d), rot, trans, cam).numpy(),
                               O.prev_d2para(d, rot, trans, cam), rtol=1e-6)


def test_training_graph_forward_equals_inference_oracle():
    """With new_traj only on frame 0 the training-mode graph (previous features / depth passed
    as arguments, m4depth_network.py:297-299) computes what the stateful inference graph does."""
    L, H, W, T = 2, 32, 48, 3
    wts = synthetic.init_weights(nbre_levels=L, seed=3, dscv_range=2, sncv_range=2, bias_std=0.05)
    samples, cam = synthetic.make_sequence(1, T, H, W, seed=11)
    ref, ref_seq = O.M4Depth(wts, nbre_levels=L, dscv_range=2, sncv_range=2)(samples, cam)
    tw = {k: torch.from_numpy(v) for k, v in wts.items()}
    tsamples = [{k: (torch.from_numpy(v) if v.dtype != np.bool_ else v) for k, v in s.items()} for s in samples]
    preds = OT.model_train(tw, tsamples, cam, L, dscv_range=2, sncv_range=2)
    for t in range(1, T):
        for lvl in range(L):
            got = preds[t][lvl]["parallax"].numpy()
            want = ref_seq[t][lvl]["parallax"]
            np.testing.assert_allclose(got, want, rtol=2e-3, atol=1e-5, err_msg=f"frame {t} level {lvl}")


def test_loss_value_and_gradients_exist():
    L, H, W, T = 2, 32, 48, 2
    wts = synthetic.init_weights(nbre_levels=L, seed=3, dscv_range=2, sncv_range=2, bias_std=0.05)
    samples, cam = synthetic.make_sequence(1, T, H, W, seed=12)
    tw = {k: torch.from_numpy(v).requires_grad_(True) for k, v in wts.items()}
    data = {k: torch.from_numpy(np.stack([s[k] for s in samples], axis=1)) for k in ("RGB_im", "depth", "rot", "trans")}
    data["new_traj"] = np.stack([s["new_traj"] for s in samples], axis=1)
    data["camera"] = cam
    loss, _ = OT.train_loss(tw, data, L, dscv_range=2, sncv_range=2)
    loss.backward()
    assert np.isfinite(loss.item()) and loss.item() > 0
    n_with_grad = sum(1 for v in tw.values() if v.grad is not None and torch.isfinite(v.grad).all() and v.grad.abs().sum() > 0)
    assert n_with_grad >= len(tw) - 2          # every conv of both levels and the encoder is reached
    for mode in ("map", "velodyne"):
        gts = [{"depth": data["depth"][:, i]} for i in range(T)]
        preds = [[{"depth": torch.full((1, H // 2, W // 2, 1), 10.0)}, {"depth": torch.full((1, H // 4, W // 4, 1), 10.0)}]] * T
        val = OT.m4depth_loss(gts, preds, mode).item()
        assert np.isfinite(val) and val > 0
