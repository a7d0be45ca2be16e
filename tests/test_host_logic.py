"""Host-side logic on CPU: the PyTorch convolution wrappers with TF 'SAME' padding,
DINL, encoder / refiner stacks and the metric classes against the oracle; synthetic
data contract; sharding arithmetic."""
import numpy as np
import pytest
import torch

from oracle import m4depth_oracle as O
from m4depth_amd import synthetic as S
from m4depth_amd import network as N
from m4depth_amd import metrics as MT
from m4depth_amd import dist as D
from helpers import F, torch_convolutions_on_cpu


@pytest.mark.parametrize("stride,h,w", [(1, 6, 7), (2, 8, 10), (2, 7, 9), (2, 6, 9)])
def test_conv_same_tf_padding(stride, h, w):
    rng = np.random.default_rng(stride * 100 + h)
    x = rng.standard_normal([2, h, w, 5]).astype(F)
    k = rng.standard_normal([3, 3, 5, 4]).astype(F)
    bias = rng.standard_normal([4]).astype(F)
    conv = N._Conv3x3SameTF(4, stride, 5)
    conv.load_hwio(k, bias, torch.device("cpu"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):        # the product layer is a HIP kernel and nothing else
        conv(torch.from_numpy(x))
    ref = O.conv2d_same(x, k, bias, stride)
    with torch_convolutions_on_cpu():
        got = conv(torch.from_numpy(x)).numpy()
        assert got.shape == ref.shape == (2, -(-h // stride), -(-w // stride), 4)
        assert np.max(np.abs(got - ref)) < 1e-5
        got = conv(torch.from_numpy(x), slope=0.1).numpy()
        assert np.max(np.abs(got - O.leaky_relu(ref, 0.1))) < 1e-5
    # load_hwio on a built layer of the same shape overwrites in place (same addresses: captured graphs stay valid)
    ptrs = (conv.weight.data_ptr(), conv.bias.data_ptr())
    conv.load_hwio(2 * k, bias + 1, torch.device("cpu"))
    assert (conv.weight.data_ptr(), conv.bias.data_ptr()) == ptrs
    assert torch.equal(conv.bias, torch.from_numpy(bias + 1)) and torch.equal(conv.weight.permute(2, 3, 1, 0), torch.from_numpy(2 * k))


def test_domain_normalization_uses_variance_not_std():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal([2, 8, 9, 6]) * 3 + 1).astype(F)
    dn = N.DomainNormalization()
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # the product layer is GPU-only
        dn(torch.from_numpy(x))
    with torch_convolutions_on_cpu():
        got = dn(torch.from_numpy(x)).numpy()
    ref = O.domain_normalization(x, np.ones(6, F), np.zeros(6, F))
    assert np.max(np.abs(got - ref)) < 1e-6
    assert np.allclose((got * got).sum(-1), 1.0, atol=1e-5)


def test_encoder_and_refiner_match_oracle():
    L = 3
    W = S.init_weights(L, seed=11, bias_std=0.1)
    model = N.M4Depth(nbre_levels=L)
    model.load_numpy_weights(W, torch.device("cpu"))
    rng = np.random.default_rng(2)
    img = rng.random([1, 32, 48, 3]).astype(F)
    lvl = model.d_estimator.levels[1]
    fin = rng.standard_normal([1, 8, 12, lvl.f_in]).astype(F)
    with torch_convolutions_on_cpu():
        got = model.encoder(torch.from_numpy(img))
        out = lvl.disp_refiner(torch.from_numpy(fin))
    ref = O.feature_pyramid(img, W, L)
    assert [tuple(g.shape) for g in got] == [(1, 16, 24, 16), (1, 8, 12, 32), (1, 4, 6, 64)]
    for g, r in zip(got, ref):
        assert np.max(np.abs(g.numpy() - r)) < 2e-5
    assert out[0].shape == (1, 8, 12, 5) and out[1].shape == (1, 8, 12, 96)        # [out5, prep96] quirk
    assert np.max(np.abs(out[0].numpy() - O.disp_refiner(fin, W, 2))) < 5e-5


def test_level_widths_and_cuts():
    assert [S.nbre_cuts_for(d) for d in range(1, 7)] == [1, 2, 2, 4, 4, 8]
    assert [S.f_input_channels(S.nbre_cuts_for(d)) for d in range(1, 7)] == [64, 122, 122, 238, 238, 470]
    assert S.f_input_channels(2, level_memory=False) == 118 and S.f_input_channels(2, SNCV=False) == 24
    assert S.f_input_channels(1, dscv_range=6, sncv_range=6) == 13 + 169 + 6
    ab = N.M4depthAblationParameters(subdivide_features=False)
    settings = {"nbre_lvls": 6, "is_training": False, "ablation": ab}
    assert N.DepthEstimatorLevel(settings, 6).nbre_cuts == 1


def test_metric_classes_match_oracle():
    rng = np.random.default_rng(3)
    gt = (80 * rng.random([2, 16, 20, 1])).astype(F)
    gt[0, :4] = 0.0
    est = (gt * (1 + 0.2 * rng.standard_normal(gt.shape)) + 0.01).astype(F)
    mets = MT.default_metrics()
    assert [m.name for m in mets] == ["AbsRel", "SqRel", "RMSE", "RMSE_log", "Delta1", "Delta2", "Delta3"]
    g = torch.clamp(torch.from_numpy(gt), 0.0, 80.0)
    e = torch.clamp(torch.from_numpy(est), 0.001, 80.0)
    for _ in range(2):                                   # Keras Mean: two identical updates, same mean
        for m in mets:
            m.update_state(g, e)
    got = np.array([float(m.result()) for m in mets])
    ref = O.metrics_batch(gt, est)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)) < 1e-5
    assert all(m.count == 2 for m in mets)
    mets[0].reset_state()
    assert mets[0].count == 0 and float(mets[0].result()) == 0.0


def test_synthetic_contract():
    samples, cam = S.make_sequence(2, 3, 32, 64, seed=5)
    assert len(samples) == 3 and samples[0]["RGB_im"].shape == (2, 32, 64, 3)
    assert samples[0]["new_traj"].all() and not samples[1]["new_traj"].any()
    assert np.allclose(np.linalg.norm(samples[1]["rot"], axis=1), 1.0, atol=1e-6)
    assert (np.linalg.norm(samples[1]["trans"], axis=1) > 1e-3).all()
    assert np.array_equal(cam["f"], np.array([[32., 16.], [32., 16.]], F))
    assert samples[0]["depth"].min() >= 1.0 and samples[0]["depth"].max() <= 80.0
    s2, _ = S.make_sequence(2, 3, 32, 64, seed=5)
    assert np.array_equal(s2[2]["RGB_im"], samples[2]["RGB_im"])
    W = S.init_weights(6)
    assert W["lvl.6.conv.0.kernel"].shape == (3, 3, 470, 128) and W["enc.s2.5.kernel"].shape == (3, 3, 192, 192)


def test_shard_range():
    assert [D.shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    with pytest.raises(ValueError):
        D.shard_range(10, 0, 4)


def test_tensor_helpers_match_oracle_on_cpu():
    """get_rot_mat / get_coords_2d / tile_in_batch are plain tensor plumbing (no kernel): the product functions themselves
    (utils/depth_operations.py:18-68, 217-221) against the oracle, bit for bit, without a GPU."""
    import m4depth_amd as M
    rng = np.random.default_rng(4)
    b, h, w = 2, 5, 6
    q = rng.standard_normal([b, 4]).astype(F)                      # NOT renormalised by the reference
    assert np.array_equal(M.get_rot_mat(torch.from_numpy(q)).numpy(), O.get_rot_mat(q))
    e = (0.02 * rng.standard_normal([b, 3])).astype(F)
    assert np.array_equal(M.get_rot_mat(torch.from_numpy(e)).numpy(), O.get_rot_mat(e))
    assert np.array_equal(M.get_rot_mat(torch.tensor([[1., 0., 0., 0.]])).numpy()[0], np.eye(3, dtype=F))   # Appendix B inv. 1
    with pytest.raises(ValueError):
        M.get_rot_mat(torch.zeros(2, 2))
    cam = {"f": np.array([[3.1, 2.2], [2.9, 2.4]], F), "c": np.array([[3.0, 2.5], [2.7, 2.6]], F)}
    coords, mesh = M.get_coords_2d(torch.zeros(b, h, w, 1), {k: torch.from_numpy(v) for k, v in cam.items()})
    oc, om = O.get_coords_2d(b, h, w, cam)
    assert coords.shape == (b, h, w, 3, 1)
    assert np.array_equal(coords.numpy()[..., 0], oc) and np.array_equal(mesh.numpy(), om)
    x = rng.standard_normal([b, 3, 2]).astype(F)
    t = M.tile_in_batch(torch.from_numpy(x), 4).numpy()
    assert t.shape == (4 * b, 3, 2) and np.array_equal(t, O.tile_in_batch(x, 4))
    assert np.array_equal(t[2 * b + 1], x[1])                       # out batch index = copy*b + bi


def test_float64_reference_mode_of_the_oracle():
    """oracle.float64_reference(): same algorithm in float64 (float16 DSCV steps kept); restores float32 afterwards."""
    rng = np.random.default_rng(8)
    b, h, w, C = 1, 6, 8, 16
    c1 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), 1)
    c2 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), 1)
    cam = {"f": np.array([[4.0, 3.0]], F), "c": np.array([[4.0, 3.0]], F)}
    rot = np.array([[1.0, 0.001, -0.002, 0.0015]], F)
    trans = np.array([[0.05, -0.02, 0.3]], F)
    disp = (0.5 + rng.random([b, h, w, 1])).astype(F)
    cv32, _ = O.get_parallax_sweeping_cv(c1, c2, disp, disp, rot, trans, cam, 2)
    with O.float64_reference():
        assert O.F32 is np.float64
        cv64, _ = O.get_parallax_sweeping_cv(c1, c2, disp, disp, rot, trans, cam, 2)
        d64 = O.parallax2depth(disp, rot, trans, cam)
    assert O.F32 is np.float32
    assert cv32.dtype == np.float32 and cv64.dtype == np.float64 and d64.dtype == np.float64
    assert np.array_equal(cv64, cv64.astype(np.float16).astype(np.float64))        # still float16-valued
    assert np.mean(cv32 == cv64) > 0.9 and np.max(np.abs(cv32 - cv64)) < 2e-3       # same up to a few float16 flips
    d32 = O.parallax2depth(disp, rot, trans, cam)
    assert d32.dtype == np.float32 and np.max(np.abs(d32 - d64) / np.abs(d64)) < 1e-5


def test_bf16_three_way_split_and_wino6_pack():
    """network_ops.split_bf16x3: every float32 value is exactly the sum of its three bf16 terms; pack_conv_weights_wino6 holds
    the same transformed filter U as the fp32 Winograd pack, in MFMA B-fragment order."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(21)
    v = np.concatenate([rng.standard_normal(4096), rng.standard_normal(512) * 1e-6, rng.standard_normal(512) * 1e6,
                        np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 255.99998, 2.0 ** -100, 3.0e38])]).astype(F)
    parts = nops.split_bf16x3(v)
    assert parts.shape == (3,) + v.shape and parts.dtype == np.uint16
    wide = (parts.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert np.array_equal(wide.sum(0), v.astype(np.float64))                         # exact, not approximately
    assert np.all(np.abs(wide[1]) <= np.abs(wide[0]) * 2.0 ** -8 + 1e-300) and np.all(np.abs(wide[2]) <= np.abs(wide[0]) * 2.0 ** -16 + 1e-300)
    cin, cout = 40, 70                                                                # padded to 48 / 128
    k = rng.standard_normal([3, 3, cin, cout]).astype(F)
    w6, cpad = nops.pack_conv_weights_wino6(k)
    assert cpad == 128 and w6.shape == (3, 2, 16, 2, 3, 64, 8)
    u = (w6.astype(np.uint32) << 16).view(np.float32).astype(np.float64).sum(4)       # [chunk, group, pos, n-tile, lane, e]
    u = u.reshape(3, 2, 16, 2, 2, 32, 8)                                              # lane = k_half * 32 + j
    full = u.transpose(2, 0, 4, 6, 1, 3, 5).reshape(16, 48, 128)                      # [pos][cin][cout]
    w8, cpad8 = nops.pack_conv_weights_winograd(k, chunk=16)                          # [chunk][pos][n][channels], same U
    ref = w8.transpose(1, 0, 3, 2).reshape(16, 48, cpad8)
    assert np.array_equal(full[:, :, :cpad8], ref.astype(np.float64)) and not full[:, :, cout:].any() and not full[:, cin:].any()


def test_small6_pack_layout():
    """pack_conv_weights_small6: [chunk][tap][CoutPad][part][16 channels] bf16, the three parts summing exactly to the weight."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(22)
    cin, cout = 36, 40                                                                # padded to 48 / 64
    k = rng.standard_normal([3, 3, cin, cout]).astype(F)
    w6, cpad = nops.pack_conv_weights_small6(k)
    assert cpad == 64 and w6.shape == (3, 9, 64, 3, 16) and w6.dtype == np.uint16
    rec = (w6.astype(np.uint32) << 16).view(np.float32).astype(np.float64).sum(3)     # [chunk, tap, n, channel]
    full = rec.transpose(1, 0, 3, 2).reshape(9, 48, 64)                               # [tap][cin][cout]
    assert np.array_equal(full[:, :cin, :cout], k.reshape(9, cin, cout).astype(np.float64))
    assert not full[:, cin:].any() and not full[:, :, cout:].any()


def test_encoder_kernel_choice_follows_the_sequence_batch(monkeypatch):
    """FeaturePyramid.set_sequence_batch: the encoder's convolutions take their kernel from the per-image grid x the SEQUENCE
    batch -- the same number in every launch mode, however many frames a mode stacks -- so a batch-32 evaluation moves the
    coarse levels (24x80 and smaller: single-small-map latency kernels at batch 1) onto the chip-filling kernels."""
    fp = N.FeaturePyramid({"ablation": N.M4depthAblationParameters(), "nbre_lvls": 6})
    convs = list(fp.conv_layers_s1) + list(fp.conv_layers_s2)
    assert all(c.per_image_dispatch and c.dispatch_batch == 1 for c in convs)
    fp.set_sequence_batch(32)
    assert all(c.dispatch_batch == 32 for c in convs)
    # level 4's stride-1 layer (24x80, 128 -> 128): small-map kernel at batch 1, the bf16-split Winograd kernel at batch 32
    assert 1 * 24 * 80 <= N.small_map_conv_pixels < 32 * 24 * 80
    assert N._use_winograd(1, 24, 80, 128, 128, 1) != 6 and N._use_winograd(32, 24, 80, 128, 128, 1) == 6
    monkeypatch.setattr(N, "encoder_batch_dispatch", False)
    fp.set_sequence_batch(32)
    assert all(c.dispatch_batch == 1 for c in convs)


def test_refiner_tail6_pack_layout():
    """pack_refiner_tail_weights6: the MFMA B fragments of the bf16-split level tail (csrc/m4d_tail6.hip).  conv6: lane
    (k-quarter kq, cout n) of tap t holds channels 8 kq .. 8 kq + 7; conv7: K-step j, k-quarter kq holds channels
    8 (kq & 1) .. + 7 of tap 2 j + (kq >> 1); the three parts sum exactly to the weight, the tenth tap and couts 5..7 are zero."""
    from m4depth_amd import network_ops as nops
    rng = np.random.default_rng(23)
    k6 = rng.standard_normal([3, 3, 32, 16]).astype(F)
    k7 = rng.standard_normal([3, 3, 16, 5]).astype(F)
    w6, w7 = nops.pack_refiner_tail_weights6(k6, k7)
    assert w6.shape == (9, 3, 64, 8) and w7.shape == (5, 3, 4, 8, 8) and w6.dtype == np.uint16 and w7.dtype == np.uint16
    widen = lambda u: (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    r6 = widen(w6).sum(1).reshape(9, 4, 16, 8)                                        # [tap, kq, n, e]
    assert np.array_equal(r6.transpose(0, 1, 3, 2).reshape(9, 32, 16), k6.reshape(9, 32, 16).astype(np.float64))
    r7 = widen(w7).sum(1).reshape(5, 2, 2, 8, 8)                                      # [j, tap parity, channel half, n, e]
    full = r7.transpose(0, 1, 2, 4, 3).reshape(10, 16, 8)                             # [tap, channel, n]
    assert np.array_equal(full[:9, :, :5], k7.reshape(9, 16, 5).astype(np.float64))
    assert not full[9].any() and not full[:, :, 5:].any()


def test_convolution_dispatch_at_the_bench_pyramid():
    """network._use_winograd at the 384x1280 / 6-level pyramid, batch 1: which kernel family every refiner layer shape gets
    (6 = bf16-split Winograd, 2 = fp32-MFMA Winograd kernel 2 / 4, 0 = direct or small-map convolution) -- the dispatch the
    measured numbers of DESIGN.md section 5 refer to; and that M4D_CONV_ARITH=f32 semantics remove kind 6 only."""
    kinds = {}
    for lvl, (h, w) in enumerate([(192, 640), (96, 320), (48, 160), (24, 80), (12, 40), (6, 20)], start=1):
        for cin, cout in [(128, 128), (128, 96), (96, 64), (64, 32)]:
            kinds[(lvl, cin, cout)] = N._use_winograd(1, h, w, cin, cout, 1)
    for lvl in (1, 2):
        assert [kinds[(lvl, ci, co)] for ci, co in [(128, 128), (128, 96), (96, 64)]] == [6, 6, 6], lvl
    assert kinds[(1, 64, 32)] == 6 and kinds[(2, 64, 32)] == 6          # 32 output channels: one HALF unit (N-tile 0 only) per tile
    # level 3 (30 tiles of 16x16): the 128-wide layers are 60 workgroups of the split kernel, 96->64 and 64->32 (half units) 30:
    # the smallest grid it takes
    assert [kinds[(3, ci, co)] for ci, co in [(128, 128), (128, 96), (96, 64), (64, 32)]] == [6, 6, 6, 6]
    assert all(kinds[(lvl, ci, co)] == 0 for lvl in (4, 5, 6) for ci, co in [(128, 128), (128, 96), (96, 64), (64, 32)])
    assert N._use_winograd(32, 24, 80, 128, 128, 1) == 6                 # batch 32: level 4 fills the chip
    assert N._use_winograd(1, 192, 640, 128, 128, 2) == 0                # stride 2 never
    before = (N.conv_arith, N.small_conv_split)
    with N.conv_arithmetic("f32"):
        assert N._use_winograd(1, 192, 640, 128, 128, 1) == 2 and N._use_winograd(1, 6, 20, 128, 128, 1) == 0
        assert N.small_conv_split is False                               # everything derived from the arithmetic follows it
    assert (N.conv_arith, N.small_conv_split) == before
    with pytest.raises(ValueError):
        with N.conv_arithmetic("fp16"):
            pass


def test_graph_autotune_only_at_small_batch():
    """``bench.py --gpus 8`` (configs[3]: 32 sequences per rank) must not run GraphedSequence's two-capture
    choice on every rank (8 x 1.4 s, a rank-dependent graph): the choice exists only where the staggered Winograd first round
    applies, batch <= 4."""
    from m4depth_amd import network as net
    assert net.GraphedSequence.wants_autotune(1, 9) == net.wino6_stagger_autotune
    assert net.GraphedSequence.wants_autotune(4, 9, autotune=True)
    assert not net.GraphedSequence.wants_autotune(32, 9) and not net.GraphedSequence.wants_autotune(32, 9, autotune=True)
    assert not net.GraphedSequence.wants_autotune(8, 9, autotune=True)
    assert not net.GraphedSequence.wants_autotune(1, 0, autotune=True) and not net.GraphedSequence.wants_autotune(1, 9, autotune=False)
