"""DepthEstimatorLevel glue kernels (m4depth_network.py:179-204, 218, 224-260)
exposed as tensor functions.  All dispatch to libm4depth_hip.so.
"""
from __future__ import annotations

import ctypes

import torch

from ._lib import lib, dptr, stream_ptr, check, as_f32


def normalize_cuts(x, nbre_cuts, out=None):
    """Per-cut tf.linalg.normalize of m4depth_network.py:179-189 (no epsilon)."""
    x = as_f32(x, "x")
    b, h, w, C = x.shape
    if C % nbre_cuts != 0:
        raise ValueError(f"nbre_cuts={nbre_cuts} does not divide the {C} feature channels")
    if out is None:
        out = torch.empty_like(x)
    check(lib.m4d_normalize_cuts(dptr(x, "x"), b, h, w, C, int(nbre_cuts), dptr(out, "out"), stream_ptr()),
          "m4d_normalize_cuts")
    return out


def resize_bilinear_v1(x, out_h, out_w, mul=1.0):
    """tf.compat.v1.image.resize_bilinear with its defaults (legacy coordinates:
    no half-pixel offset), m4depth_network.py:202-204."""
    x = as_f32(x, "x")
    b, ih, iw, c = x.shape
    out = torch.empty((b, out_h, out_w, c), dtype=torch.float32, device=x.device)
    check(lib.m4d_resize_bilinear_v1(dptr(x, "x"), b, ih, iw, c, int(out_h), int(out_w), float(mul), dptr(out),
                                     stream_ptr()), "m4d_resize_bilinear_v1")
    return out


def resize_nearest(x, out_h, out_w):
    """tf.image.resize(method=NEAREST_NEIGHBOR), m4depth_network.py:368."""
    x = as_f32(x, "x")
    b, ih, iw, c = x.shape
    out = torch.empty((b, out_h, out_w, c), dtype=torch.float32, device=x.device)
    check(lib.m4d_resize_nearest(dptr(x, "x"), b, ih, iw, c, int(out_h), int(out_w), dptr(out), stream_ptr()),
          "m4d_resize_nearest")
    return out


def level_pre(prev_l_est, depth_prev_t, trans, camera, b, h, w, device, f_input=None, log_off=0,
              other_off=-1, log_scale=1.0, normalize=None, depth_state_reset=None):
    """Fused upsample of the coarser level's estimate (+ prev_d2para of the
    temporal depth state, + the log-parallax / level-memory features written
    straight into ``f_input``).  Returns (para_prev_l, depth_prev_l,
    other_prev_l, para_prev_t or None).  ``normalize`` = (raw features [b,h,w,C], cuts, out): the per-cut
    normalisation of the level's current features rides in the same launch (m4d_level_pre_normalize).
    ``depth_state_reset`` [b,h,w,1] (reset branch, m4depth_network.py:209) is set to 1000 in the same pass."""
    para = torch.empty((b, h, w, 1), dtype=torch.float32, device=device)
    depth = torch.empty_like(para)
    other = torch.empty((b, h, w, 4), dtype=torch.float32, device=device)
    para_t = torch.empty_like(para) if depth_prev_t is not None else None
    if prev_l_est is not None:
        pd = as_f32(prev_l_est["depth"], "prev_l_est['depth']")
        pp = as_f32(prev_l_est["parallax"], "prev_l_est['parallax']")
        po = as_f32(prev_l_est["other"], "prev_l_est['other']")
        ph, pw = pd.shape[1:3]
    else:
        pd = pp = po = None
        ph = pw = 0
    tr = f = c = None
    if depth_prev_t is not None:
        tr = as_f32(trans, "trans").reshape(b, 3)
        f = as_f32(camera["f"], "camera['f']").reshape(b, 2)
        c = as_f32(camera["c"], "camera['c']").reshape(b, 2)
    f_stride = f_input.shape[-1] if f_input is not None else 0
    if normalize is not None:
        raw, cuts, nout = normalize
        raw = as_f32(raw, "curr_f_maps")
        check(lib.m4d_level_pre_normalize(dptr(pd), dptr(pp), dptr(po), ph, pw, dptr(depth_prev_t, "depth_prev_t"), dptr(tr),
                                          dptr(f), dptr(c), b, h, w, dptr(para), dptr(depth), dptr(other), dptr(para_t),
                                          dptr(f_input, "f_input"), f_stride, int(log_off), int(other_off), float(log_scale),
                                          dptr(depth_state_reset, "depth_state_reset"), dptr(raw, "curr_f_maps"), int(raw.shape[-1]), int(cuts), dptr(nout, "normalised features"),
                                          stream_ptr()), "m4d_level_pre_normalize")
        return para, depth, other, para_t
    check(lib.m4d_level_pre(dptr(pd), dptr(pp), dptr(po), ph, pw, dptr(depth_prev_t, "depth_prev_t"), dptr(tr),
                            dptr(f), dptr(c), b, h, w, dptr(para), dptr(depth), dptr(other), dptr(para_t),
                            dptr(f_input, "f_input"), f_stride, int(log_off), int(other_off), float(log_scale),
                            dptr(depth_state_reset, "depth_state_reset"), stream_ptr()), "m4d_level_pre")
    return para, depth, other, para_t


def pyramid_reset(levels, b):
    """The reset frame of every level in one launch (m4d_pyramid_reset).  ``levels`` = coarse -> fine list of dicts
    ``features`` [b,h,w,C] raw, ``cuts``, ``state_features`` (receives the normalised features), ``depth_state`` [b,h,w,1].
    Returns the per-level estimates (coarse -> fine) ``{"depth", "parallax", "other"}``: the constant maps the x2 upsampling
    chain of m4depth_network.py:198-204 produces on a new trajectory."""
    from ._lib import ResetLevel
    n = len(levels)
    arr = (ResetLevel * n)()
    ests = []
    pv = 1.0
    for i, lv in enumerate(levels):
        f = as_f32(lv["features"], "curr_f_maps")
        bb, h, w, c = f.shape
        if bb != b:
            raise ValueError("pyramid_reset: batch mismatch")
        para = torch.empty((b, h, w, 1), dtype=torch.float32, device=f.device)
        depth = torch.empty_like(para)
        other = torch.empty((b, h, w, 4), dtype=torch.float32, device=f.device)
        arr[i] = ResetLevel(dptr(f, "curr_f_maps").value, dptr(lv["state_features"], "state_features").value,
                            dptr(lv["depth_state"], "depth_state").value, dptr(para).value, dptr(depth).value, dptr(other).value,
                            h, w, c, int(lv["cuts"]), pv)
        ests.append({"depth": depth, "parallax": para, "other": other})
        pv *= 2.0                                    # :203, exact
    check(lib.m4d_pyramid_reset(arr, n, int(b), stream_ptr()), "m4d_pyramid_reset")
    return ests


def normalize_levels(maps):
    """The per-cut normalisation of several feature maps in ONE launch (m4d_normalize_levels; the bits of ``normalize_cuts``).
    ``maps`` = [(tensor [n,h,w,C], cuts), ...] (at most 8); returns the normalised tensors in the same order."""
    from ._lib import NormLevel
    n = len(maps)
    arr = (NormLevel * n)()
    outs = []
    for i, (x, cuts) in enumerate(maps):
        x = as_f32(x, "features")
        out = torch.empty_like(x)
        arr[i] = NormLevel(dptr(x, "features").value, dptr(out).value, int(x.shape[0] * x.shape[1] * x.shape[2]), int(x.shape[3]), int(cuts))
        outs.append(out)
    check(lib.m4d_normalize_levels(arr, n, stream_ptr()), "m4d_normalize_levels")
    return outs


def level_post(refiner_out, rot, trans, camera, scale, depth_state=None):
    """Fused tail of a level (m4depth_network.py:247-260): returns (parallax,
    depth, other); ``depth_state`` (optional) receives the depth as well."""
    ro = as_f32(refiner_out, "refiner_out")
    b, h, w, five = ro.shape
    if five != 5:
        raise ValueError(f"refiner output must have 5 channels, got {five}")
    rot = as_f32(rot, "rot")
    tr = as_f32(trans, "trans").reshape(b, 3)
    f = as_f32(camera["f"], "camera['f']").reshape(b, 2)
    c = as_f32(camera["c"], "camera['c']").reshape(b, 2)
    para = torch.empty((b, h, w, 1), dtype=torch.float32, device=ro.device)
    depth = torch.empty_like(para)
    other = torch.empty((b, h, w, 4), dtype=torch.float32, device=ro.device)
    check(lib.m4d_level_post(dptr(ro), dptr(rot, "rot"), rot.shape[1], dptr(tr), dptr(f), dptr(c), b, h, w,
                             float(scale), dptr(para), dptr(depth), dptr(other), dptr(depth_state, "depth_state"),
                             stream_ptr()), "m4d_level_post")
    return para, depth, other


def bias_act_(x, bias, slope=0.1):
    """In-place convolution epilogue on an NHWC tensor: x = leaky_relu(x + bias, slope)
    (slope = 1.0: bias add only)."""
    C = x.shape[-1]
    rows = x.numel() // C
    check(lib.m4d_bias_act(dptr(x, "x"), dptr(bias, "bias"), rows, C, float(slope), dptr(x, "x"), stream_ptr()),
          "m4d_bias_act")
    return x


_ws_cache = {}


def _workspace(key, nbytes, device):
    """Scratch buffer per (purpose, device, stream): kernels of different streams may run
    concurrently (the level pipeline of DepthEstimatorPyramid), so they must not share scratch."""
    k = (key, device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _ws_cache.get(k)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _ws_cache[k] = ws
    return ws


_zero_ws_cache = {}


def zeroed_workspace(key, shape, device):
    """A persistent float32 buffer of ``shape`` per (purpose, shape, device, stream), zero-filled ONCE when it is created:
    for buffers whose padding elements are never written (the channel-padded refiner input).  Created outside any
    hipGraph capture by the eager warm-up pass; replays only reuse it."""
    k = (key, tuple(shape), device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _zero_ws_cache.get(k)
    if buf is None:
        buf = torch.zeros(tuple(shape), dtype=torch.float32, device=device)
        _zero_ws_cache[k] = buf
    return buf


def dinl_act(x, scale, bias, slope=1.0, out=None, offset=(0, 0)):
    """DomainNormalization (m4depth_network.py:44-48) fused with leaky_relu(slope)
    (slope = 1.0: normalisation alone).  x [b,h,w,C], C in (16, 32).  ``out`` may be a larger
    zero-bordered [b,H',W',C] buffer; the result is then written at ``offset`` (row, col)."""
    x = as_f32(x, "x")
    b, h, w, C = x.shape
    ws = _workspace("dinl", 4 * int(lib.m4d_dinl_workspace_floats(b, C)), x.device)
    if out is None:
        out = torch.empty_like(x)
    check(lib.m4d_dinl_fwd_padded(dptr(x, "x"), dptr(scale.reshape(-1), "scale"), dptr(bias.reshape(-1), "bias"),
                                  b, h, w, C, float(slope), dptr(ws), dptr(out, "out"), out.shape[1], out.shape[2],
                                  int(offset[0]), int(offset[1]), stream_ptr()), "m4d_dinl_fwd_padded")
    return out


def bias_act_padded(x, bias, slope, out, offset=(0, 0)):
    """leaky_relu(x + bias, slope) written into the zero-bordered buffer ``out`` [b,H',W',C] at
    ``offset``: the activation arrives already TF-'SAME'-padded for the stride-2 convolution."""
    b, h, w, C = x.shape
    check(lib.m4d_bias_act_padded(dptr(x, "x"), dptr(bias, "bias"), b, h, w, C, float(slope), dptr(out, "out"),
                                  out.shape[1], out.shape[2], int(offset[0]), int(offset[1]), stream_ptr()),
          "m4d_bias_act_padded")
    return out


def depth_metrics(gt, est, max_d=80.0, total=None, count=0, mean=None):
    """The 7 metrics of metrics.py for one batch, one pass: returns a [7] device tensor
    (AbsRel, SqRel, RMSE, RMSE_log, Delta1, Delta2, Delta3); clipping as in test_step.
    ``total`` [7] (Keras-Mean totals) is incremented and ``mean`` [7] = total / count written in the same launch."""
    est = as_f32(est, "est")
    if gt.numel() != est.numel():
        raise ValueError(f"gt {tuple(gt.shape)} and est {tuple(est.shape)} differ in size")
    ws = _workspace("metrics", int(lib.m4d_metrics_workspace_bytes()), est.device)
    out = torch.empty(7, dtype=torch.float32, device=est.device)
    if (gt.is_cuda and gt.dtype == torch.float32 and gt.dim() >= 2 and not gt.is_contiguous() and gt[0].is_contiguous()
            and gt.stride(0) >= gt[0].numel()):
        # dense images at a batch stride: data["depth"][:, -1] of a [b,T,H,W,1] sequence batch, read in place
        check(lib.m4d_depth_metrics_strided(ctypes.c_void_p(gt.data_ptr()), gt[0].numel(), gt.stride(0), dptr(est, "est"),
                                            gt.numel(), float(max_d), dptr(ws), dptr(out), dptr(total, "total"), float(count),
                                            dptr(mean, "mean"), stream_ptr()), "m4d_depth_metrics_strided")
        return out
    gt = as_f32(gt, "gt")
    check(lib.m4d_depth_metrics(dptr(gt, "gt"), dptr(est, "est"), gt.numel(), float(max_d), dptr(ws), dptr(out),
                                dptr(total, "total"), float(count), dptr(mean, "mean"), stream_ptr()), "m4d_depth_metrics")
    return out


def pack_conv_weights(kernel_hwio):
    """Re-pack a TF HWIO [3,3,Cin,Cout] kernel for m4d_conv3x3_bias_act:
    [ceil(Cin/16)][9][CoutPad][16], CoutPad = Cout rounded up to 32, the 16 input channels of a
    chunk stored even-first, zero padded.  numpy in, numpy out (host side, once per model)."""
    import numpy as np
    k = np.asarray(kernel_hwio, dtype=np.float32)
    assert k.shape[:2] == (3, 3)
    cin, cout = k.shape[2], k.shape[3]
    nch = -(-cin // 16)
    cpad = -(-cout // 32) * 32
    full = np.zeros((3, 3, nch * 16, cpad), np.float32)
    full[:, :, :cin, :cout] = k
    perm = np.array([0, 2, 4, 6, 8, 10, 12, 14, 1, 3, 5, 7, 9, 11, 13, 15])
    w = full.reshape(9, nch, 16, cpad)[:, :, perm, :]            # [tap][chunk][16p][n]
    return np.ascontiguousarray(w.transpose(1, 0, 3, 2)), cpad    # [chunk][tap][n][16p]


def conv3x3_bias_act(x, wp, bias, cout, cout_pad, slope=0.1, stride=1):
    """3x3 TF-'SAME' convolution (stride 1 or 2) + bias + leaky_relu(slope) on the matrix cores."""
    x = as_f32(x, "x")
    b, h, w, cin = x.shape
    oh, ow = -(-h // stride), -(-w // stride)
    out = torch.empty((b, oh, ow, cout), dtype=torch.float32, device=x.device)
    ws = None
    ws_floats = 0
    if b * oh * ow <= 16384:                             # coarse levels: allow split-K through a workspace
        ws_floats = int(lib.m4d_conv3x3_workspace_floats(b, oh, ow, int(cout_pad)))
        ws = _workspace("conv_splitk", 4 * ws_floats, x.device)
    check(lib.m4d_conv3x3s_bias_act_ws(dptr(x, "x"), dptr(wp, "wp"), dptr(bias, "bias"), b, h, w, cin, int(cout),
                                       int(cout_pad), int(stride), float(slope), dptr(out), dptr(ws), ws_floats,
                                       stream_ptr()), "m4d_conv3x3s_bias_act_ws")
    return out


def conv3x3_small_bias_act(x, wp, bias, cout, cout_pad, slope=0.1, stride=1):
    """The convolution for small maps (stride 1 or 2): one launch, K split over the waves of a workgroup (no split-K reduce)."""
    x = as_f32(x, "x")
    b, h, w, cin = x.shape
    out = torch.empty((b, -(-h // stride), -(-w // stride), cout), dtype=torch.float32, device=x.device)
    check(lib.m4d_conv3x3s_small_bias_act(dptr(x, "x"), dptr(wp, "wp"), dptr(bias, "bias"), b, h, w, cin, int(cout),
                                          int(cout_pad), int(stride), float(slope), dptr(out), stream_ptr()),
          "m4d_conv3x3s_small_bias_act")
    return out


_WINO_G = [[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]]


_WINO_SKEW = 64


def pack_conv_weights_winograd(kernel_hwio, chunk=16):
    """Winograd F(2x2,3x3) filter transform U = G g G^T of a TF HWIO [3,3,Cin,Cout] kernel (in float64, rounded
    once to float32), packed for m4d_conv3x3_wino_bias_act (chunk = 16) / m4d_conv3x3_wino2_bias_act (chunk = 8):
    [ceil(Cin/chunk)][16 positions][CoutPad][chunk channels], zero padded; for chunk = 8 every position block is
    followed by 64 floats of padding (kWSkew in m4d_wino.hip: the 16 position blocks of a chunk, 4 KB apart at
    CoutPad = 128, otherwise all start on the same L2 channel).  numpy in, numpy out."""
    import numpy as np
    k = np.asarray(kernel_hwio, dtype=np.float64)
    assert k.shape[:2] == (3, 3)
    cin, cout = k.shape[2], k.shape[3]
    G = np.array(_WINO_G, np.float64)
    U = np.einsum('ij,jkco,lk->ilco', G, k, G)                   # [4,4,Cin,Cout]
    nch = -(-cin // chunk)
    cpad = -(-cout // 32) * 32
    full = np.zeros((16, nch * chunk, cpad), np.float32)
    full[:, :cin, :cout] = U.reshape(16, cin, cout).astype(np.float32)
    w = np.ascontiguousarray(full.reshape(16, nch, chunk, cpad).transpose(1, 0, 3, 2))    # [chunk][pos][n][channels]
    if chunk == 8:
        skewed = np.zeros((nch, 16, cpad * chunk + _WINO_SKEW), np.float32)
        skewed[:, :, :cpad * chunk] = w.reshape(nch, 16, cpad * chunk)
        return skewed, cpad
    return w, cpad


def conv3x3_wino_bias_act(x, wu, bias, cout, cout_pad, slope=0.1):
    """3x3 stride-1 TF-'SAME' convolution + bias + leaky_relu(slope), Winograd F(2x2,3x3) on the matrix cores."""
    x = as_f32(x, "x")
    b, h, w, cin = x.shape
    out = torch.empty((b, h, w, cout), dtype=torch.float32, device=x.device)
    check(lib.m4d_conv3x3_wino_bias_act(dptr(x, "x"), dptr(wu, "wu"), dptr(bias, "bias"), b, h, w, cin, int(cout),
                                        int(cout_pad), float(slope), dptr(out), stream_ptr()), "m4d_conv3x3_wino_bias_act")
    return out


def conv3x3_wino2_bias_act(x, wu8, bias, cout, cout_pad, slope=0.1):
    """Winograd variant 2 (16x16 workgroup tile, 8-channel chunks; weights packed with chunk=8)."""
    x = as_f32(x, "x")
    b, h, w, cin = x.shape
    out = torch.empty((b, h, w, cout), dtype=torch.float32, device=x.device)
    check(lib.m4d_conv3x3_wino2_bias_act(dptr(x, "x"), dptr(wu8, "wu8"), dptr(bias, "bias"), b, h, w, cin, int(cout),
                                         int(cout_pad), float(slope), dptr(out), stream_ptr()), "m4d_conv3x3_wino2_bias_act")
    return out


def split_bf16x3(v):
    """float32 array -> uint16 [3, ...]: three bf16 bit patterns p1, p2, p3 (round to nearest even) with
    v == p1 + p2 + p3 EXACTLY (8 + 8 + 8 significand bits; the residuals of a rounding are exact in float32)."""
    import numpy as np

    def bf16_rn(x):
        u = x.view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)

    def widen(b):
        return (b.astype(np.uint32) << 16).view(np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    if not np.isfinite(v).all():
        raise ValueError("split_bf16x3: non-finite float32 value (NaN / Inf weights cannot be split into bf16 terms)")
    # below 2^-110 the third residual would fall into bf16's denormal range and stop being exact: such magnitudes are
    # 30 orders below anything a float32 accumulation of activations x weights can resolve -- flushed to zero
    v = np.where(np.abs(v) < np.float32(2.0 ** -110), np.float32(0.0), v)
    p1 = bf16_rn(v)
    r1 = v - widen(p1)
    p2 = bf16_rn(r1)
    r2 = r1 - widen(p2)
    p3 = bf16_rn(r2)
    if not np.array_equal(widen(p3), r2):
        raise ValueError("split_bf16x3: a float32 value is not the exact sum of three bf16 terms")
    return np.stack([p1, p2, p3])


def pack_conv_weights_wino6(kernel_hwio):
    """Winograd F(2x2,3x3) filter transform U = G g G^T (float64, rounded once to float32 -- the same U as
    pack_conv_weights_winograd), every value split exactly into three bf16 terms, in the MFMA B-fragment order of
    m4d_conv3x3_wino6_bias_act: [Cin/16][CoutPad/64][16 positions][2 N-tiles][3 parts][64 lanes][8] bf16 (uint16 bits),
    lane = k_half * 32 + cout % 32, element e = channel 8 k_half + e of the chunk.  Cin is zero-padded to a multiple of
    16, Cout to a multiple of 64.  numpy in, (numpy uint16, CoutPad) out."""
    import numpy as np
    k = np.asarray(kernel_hwio, dtype=np.float64)
    assert k.shape[:2] == (3, 3)
    cin, cout = k.shape[2], k.shape[3]
    G = np.array(_WINO_G, np.float64)
    U = np.einsum('ij,jkco,lk->ilco', G, k, G)                   # [4,4,Cin,Cout]
    nch, ng = -(-cin // 16), -(-cout // 64)
    cpad = ng * 64
    full = np.zeros((16, nch * 16, cpad), np.float32)
    full[:, :cin, :cout] = U.reshape(16, cin, cout).astype(np.float32)
    parts = split_bf16x3(full)                                   # [3][pos][cin][cout]
    parts = parts.reshape(3, 16, nch, 2, 8, ng, 2, 32)           # part, pos, chunk, k_half, e, n-group, n-tile, j
    w = np.ascontiguousarray(parts.transpose(2, 5, 1, 6, 0, 3, 7, 4))   # chunk, group, pos, n-tile, part, k_half, j, e
    return w.reshape(nch, ng, 16, 2, 3, 64, 8), cpad


def pack_conv_weights_small6(kernel_hwio):
    """A TF HWIO [3,3,Cin,Cout] kernel split exactly into three bf16 terms per weight, for m4d_conv3x3_small6_bias_act:
    [ceil(Cin/16)][9 taps][CoutPad][3 parts][16 channels] bf16 (uint16 bits), Cin zero-padded to a multiple of 16, CoutPad = Cout
    rounded up to 32.  numpy in, (numpy uint16, CoutPad) out."""
    import numpy as np
    k = np.asarray(kernel_hwio, dtype=np.float32)
    assert k.shape[:2] == (3, 3)
    cin, cout = k.shape[2], k.shape[3]
    nch = -(-cin // 16)
    cpad = -(-cout // 32) * 32
    full = np.zeros((9, nch * 16, cpad), np.float32)
    full[:, :cin, :cout] = k.reshape(9, cin, cout)
    parts = split_bf16x3(full).reshape(3, 9, nch, 16, cpad)       # part, tap, chunk, channel, n
    return np.ascontiguousarray(parts.transpose(2, 1, 4, 0, 3)), cpad   # chunk, tap, n, part, channel


def pack_conv_weights_lat(kernel_hwio):
    """A TF HWIO [3,3,Cin,Cout] kernel split exactly into three bf16 terms per weight, in the MFMA B-fragment order
    m4d_conv3x3_lat loads straight into operand registers: [ceil(Cout/32)][ceil(Cin/16)][9 taps][3 parts][64 lanes][8] bf16
    (uint16 bits); lane = k_half * 32 + cout % 32, element e = channel 8 k_half + e of the chunk.  Zero padded.
    numpy in, numpy uint16 out."""
    import numpy as np
    k = np.asarray(kernel_hwio, dtype=np.float32)
    assert k.shape[:2] == (3, 3)
    cin, cout = k.shape[2], k.shape[3]
    nch, ng = -(-cin // 16), -(-cout // 32)
    full = np.zeros((9, nch * 16, ng * 32), np.float32)
    full[:, :cin, :cout] = k.reshape(9, cin, cout)
    parts = split_bf16x3(full).reshape(3, 9, nch, 2, 8, ng, 32)   # part, tap, chunk, k_half, e, group, n
    return np.ascontiguousarray(parts.transpose(5, 2, 1, 0, 3, 6, 4)).reshape(ng, nch, 9, 3, 64, 8)


class PartialAct:
    """An activation that exists as K-slice partial sums: ``slabs`` [S,b,h,w,C] raw sums of m4d_conv3x3_lat's K slices, to be
    added in slab order, + ``bias``, leaky_relu(``slope``) -- which the consuming m4d_conv3x3_lat does while it stages its
    input.  ``dense()`` = the finished [b,h,w,C] tensor (one HIP launch, m4d_partial_finish: same order, same bits)."""

    def __init__(self, slabs, bias, slope):
        self.slabs, self.bias, self.slope = slabs, bias, 1.0 if slope is None else float(slope)

    @property
    def shape(self):
        return self.slabs.shape[1:]

    @property
    def is_cuda(self):
        return self.slabs.is_cuda

    @property
    def device(self):
        return self.slabs.device

    def dense(self):
        S, b, h, w, C = self.slabs.shape
        out = torch.empty((b, h, w, C), dtype=torch.float32, device=self.slabs.device)
        check(lib.m4d_partial_finish(dptr(self.slabs, "slabs"), S, b * h * w * C, dptr(self.bias, "bias"), self.slope,
                                     b * h * w, C, dptr(out), stream_ptr()), "m4d_partial_finish")
        return out


def _lat_halo_pixels(mt, stride):
    mtx, mty = (2 if mt >= 4 else 1), (2 if mt >= 2 else 1)
    return ((8 * mtx - 1) * stride + 3) * ((4 * mty - 1) * stride + 3)


import os as _os
# grid cap of the latency-first small-map convolution (workgroups): 256 = one per CU
lat_max_workgroups = int(_os.environ.get("M4D_LAT_MAX_WG", "256"))


def lat_config(b, h, w, cin, cout, final=False, stride=1, mw=False):
    """(mt, kw, s_out) of m4d_conv3x3_lat / m4d_conv3x3s_lat for a layer -- the rule the sweep of tools/bench_lat_convs.py
    reads out (profiles/r05_lat_conv_sweep.txt; ``mw``: the M-over-waves form for narrow short-K layers on larger maps, see
    csrc/m4d_convlat.hip): as many K slices over workgroups as there can be (s_out <= 4 partial slabs, no
    empty slice; ``final`` forces 1: the consumer cannot add slabs), one chunk per wave where the slice allows it (kw = 1 / 2 /
    4 K sub-slices over the waves of a workgroup), and a grid of at most one workgroup per CU: on larger maps first the waves
    stop splitting K (kw -> 1: four cout groups share one staged halo), then a wave takes 2 / 4 M-tiles.  (h, w) = the INPUT
    size; two staged rounds of kw chunks must fit the 160 KB of LDS."""
    n_chunks, n_groups = -(-cin // 16), -(-cout // 32)
    oh, ow = -(-h // stride), -(-w // stride)
    if mw:                                              # "M over waves": mt code 8, one cout group per workgroup, K unsplit
        return 8, 1, 1
    s_out = 1 if final else min(4, n_chunks)
    while s_out > 1 and (s_out - 1) * (-(-n_chunks // s_out)) >= n_chunks:
        s_out -= 1                                      # no empty K slice
    cps = -(-n_chunks // s_out)
    kw = 1 if cps <= 1 else (2 if cps <= 2 else 4)

    def wgs(mt, kw):
        mtx, mty = (2 if mt == 4 else 1), (2 if mt >= 2 else 1)
        tiles = b * (-(-oh // (4 * mty))) * (-(-ow // (8 * mtx)))
        return tiles * (-(-n_groups // (4 // kw))) * s_out

    def fits(mt, kw):
        return 2 * kw * _lat_halo_pixels(mt, stride) * 96 <= 160 * 1024
    mt = 1
    while (wgs(mt, kw) > lat_max_workgroups or not fits(mt, kw)) and kw > 1:
        kw //= 2
    while wgs(mt, kw) > lat_max_workgroups and mt < 4 and fits(mt * 2, kw):
        mt *= 2
    return mt, kw, s_out


def conv3x3_lat(x, wp, bias, cout, slope=0.1, final=False, config=None, stride=1):
    """m4d_conv3x3_lat / m4d_conv3x3s_lat: ``x`` a finished [b,h,w,Cin] tensor or a ``PartialAct``; returns a finished tensor
    (s_out == 1) or a ``PartialAct`` the next call finishes while staging.  ``config`` = (mt, kw, s_out) overrides lat_config;
    ``stride`` 1 or 2 (TF 'SAME')."""
    if isinstance(x, PartialAct):
        xs, s_in, x_bias, x_slope = x.slabs, x.slabs.shape[0], x.bias, x.slope
        b, h, w, cin = x.shape
        slab = b * h * w * cin
    else:
        xs = as_f32(x, "x")
        s_in, x_bias, x_slope = 1, None, 1.0
        b, h, w, cin = xs.shape
        slab = 0
    mt, kw, s_out = config if config is not None else lat_config(b, h, w, cin, cout, final, stride)
    act = 1.0 if slope is None else float(slope)
    oh, ow = -(-h // stride), -(-w // stride)
    out = torch.empty((s_out, b, oh, ow, cout), dtype=torch.float32, device=xs.device)
    check(lib.m4d_conv3x3s_lat(dptr(xs, "x"), s_in, slab, dptr(x_bias, "x_bias"), float(x_slope), dptr(wp, "wp", torch.int16),
                               dptr(bias, "bias"), b, h, w, cin, int(cout), int(stride), act, mt, kw, s_out, dptr(out),
                               b * oh * ow * int(cout), stream_ptr()), "m4d_conv3x3s_lat")
    return out[0] if s_out == 1 else PartialAct(out, bias, act)


def conv3x3_small6_bias_act(x, wp6, bias, cout, cout_pad, slope=0.1):
    """The one-launch small-map convolution with float32 operands split into three bf16 terms (csrc/m4d_conv.hip,
    conv3x3_small6_kernel): float32 accuracy at 2.7x less matrix-core time per wave."""
    x = as_f32(x, "x")
    b, h, w, cin = x.shape
    out = torch.empty((b, h, w, cout), dtype=torch.float32, device=x.device)
    check(lib.m4d_conv3x3_small6_bias_act(dptr(x, "x"), dptr(wp6, "wp6", torch.int16), dptr(bias, "bias"), b, h, w, cin, int(cout),
                                          int(cout_pad), float(slope), dptr(out), stream_ptr()), "m4d_conv3x3_small6_bias_act")
    return out


def conv3x3_wino6_bias_act(x, wu6, bias, cout, cout_pad, slope=0.1, kernel=0, stagger_us=0, stagger_phases=16):
    """3x3 stride-1 TF-'SAME' convolution + bias + leaky_relu(slope): Winograd F(2x2,3x3), float32 operands split into
    three bf16 terms, six bf16 MFMA products per term pair, float32 accumulation.  ``kernel``: 0 = chosen from the grid,
    1 = one workgroup per (tile, 64 couts) (csrc/m4d_wino6.hip), 2 = persistent workgroups (csrc/m4d_wino6p.hip); same bits.
    ``stagger_us`` > 0: the one-workgroup-per-unit kernel starts its first round of workgroups in ``stagger_phases`` groups spread
    over that many microseconds (a per-launch ARGUMENT, m4d_conv3x3_wino6_bias_act_ks; same bits)."""
    x = as_f32(x, "x")
    b, h, w, cin = x.shape
    out = torch.empty((b, h, w, cout), dtype=torch.float32, device=x.device)
    check(lib.m4d_conv3x3_wino6_bias_act_ks(dptr(x, "x"), dptr(wu6, "wu6", torch.int16), dptr(bias, "bias"), b, h, w, cin,
                                            int(cout), int(cout_pad), float(slope), dptr(out), int(kernel), int(stagger_us),
                                            int(stagger_phases), stream_ptr()),
          "m4d_conv3x3_wino6_bias_act_ks")
    return out


class FrameStack:
    """``frames`` [bsz,T',H,W,3]-shaped VIEW into a sequence batch [bsz,T,H,W,3] (frames t0..t1 of every sequence): what
    M4Depth.call hands the encoder so that all frames are encoded in one launch, frame-major (image t*bsz + i), read in
    place.  Only the fused encoder head understands it; ``dense()`` materialises the stacked batch for any other layer."""

    def __init__(self, frames):
        if frames.dim() != 5 or frames.dtype != torch.float32 or not frames[0, 0].is_contiguous():
            raise ValueError("FrameStack: expected a float32 [b,T,H,W,3] view with dense frames")
        self.frames = frames

    @property
    def shape(self):
        bsz, t, h, w, c = self.frames.shape
        return (bsz * t, h, w, c)

    @property
    def is_cuda(self):
        return self.frames.is_cuda

    @property
    def device(self):
        return self.frames.device

    def dense(self):
        bsz, t, h, w, c = self.frames.shape
        return self.frames.transpose(0, 1).reshape(bsz * t, h, w, c)


def encoder_head(images, w_hwio, bias1, dn_scale, dn_bias, wp2, bias2, cout2, cout2_pad, slope=0.1):
    """Encoder level 0 with DINL (m4depth_network.py:79-87) in two fused calls: conv3x3(3->16) + bias + DINL statistics,
    then the stride-2 convolution reading the raw map and normalising it on the fly.  Returns [b, h/2, w/2, cout2].
    ``images``: a dense [b,h,w,3] tensor or a ``FrameStack``."""
    if isinstance(images, FrameStack):
        fr = images.frames
        bsz = fr.shape[0]
        stride_b, stride_t = fr.stride(0), fr.stride(1)
        b, h, w, c3 = images.shape
        img_ptr = ctypes.c_void_p(fr.data_ptr())
        if not fr.is_cuda:
            raise RuntimeError("images: m4depth_amd ops run on the MI355X only; there is no CPU fallback")
        images_dev = fr.device
    else:
        images = as_f32(images, "images")
        b, h, w, c3 = images.shape
        bsz, stride_b, stride_t = b, h * w * c3, 0
        img_ptr = dptr(images, "images")
        images_dev = images.device
    if c3 != 3:
        raise ValueError(f"encoder_head expects RGB images, got {c3} channels")
    C = bias1.numel()
    ws = _workspace("dinl", 4 * int(lib.m4d_dinl_workspace_floats(b, C)), images_dev)
    raw = torch.empty((b, h, w, C), dtype=torch.float32, device=images_dev)
    check(lib.m4d_enc_head_fwd(img_ptr, int(bsz), int(stride_b), int(stride_t), dptr(w_hwio, "w_hwio"), dptr(bias1, "bias"),
                               b, h, w, C, dptr(ws), dptr(raw), stream_ptr()), "m4d_enc_head_fwd")
    total = int(lib.m4d_dinl_workspace_floats(b, C))     # [partials | mean b*C | var b*C]
    mean = ws[total - 2 * b * C: total - b * C]
    var = ws[total - b * C: total]
    oh, ow = -(-h // 2), -(-w // 2)
    out = torch.empty((b, oh, ow, cout2), dtype=torch.float32, device=images_dev)
    check(lib.m4d_conv3x3s2_dinl_bias_act(dptr(raw), dptr(mean), dptr(var), dptr(dn_scale.reshape(-1), "dn_scale"),
                                          dptr(dn_bias.reshape(-1), "dn_bias"), float(slope), dptr(wp2, "wp"), dptr(bias2, "bias"),
                                          b, h, w, int(cout2), int(cout2_pad), float(slope), dptr(out), stream_ptr()),
          "m4d_conv3x3s2_dinl_bias_act")
    return out


def _image_batch_address(images):
    """(pointer, bsz, stride_b, stride_t, (b, h, w, c), device) of a dense [b,h,w,3] batch or a ``FrameStack`` read in place."""
    if isinstance(images, FrameStack):
        fr = images.frames
        if not fr.is_cuda:
            raise RuntimeError("images: m4depth_amd ops run on the MI355X only; there is no CPU fallback")
        return ctypes.c_void_p(fr.data_ptr()), fr.shape[0], fr.stride(0), fr.stride(1), tuple(images.shape), fr.device
    images = as_f32(images, "images")
    b, h, w, c3 = images.shape
    return dptr(images, "images"), b, h * w * c3, 0, (b, h, w, c3), images.device


def encoder_level0_stats(images, w1_hwio, bias1):
    """The DINL statistics of encoder level 0 (m4depth_network.py:44-48 on the 3 -> 16 convolution's output, which is never
    written): (mean, var), each [b,16], per image.  ``images`` may hold MORE frames than the ``encoder_level0`` call that
    consumes a slice of the rows: the statistics of a whole sequence ride in the first encoder batch's launches."""
    img_ptr, bsz, stride_b, stride_t, (b, h, w, c3), dev = _image_batch_address(images)
    if c3 != 3 or bias1.numel() != 16:
        raise ValueError("encoder_level0_stats expects RGB images and a 3->16 convolution")
    ws = _workspace("dinl", 4 * int(lib.m4d_dinl_workspace_floats(b, 16)), dev)
    mean = torch.empty((b, 16), dtype=torch.float32, device=dev)
    var = torch.empty((b, 16), dtype=torch.float32, device=dev)
    check(lib.m4d_enc_level0_stats(img_ptr, int(bsz), int(stride_b), int(stride_t), dptr(w1_hwio, "w1_hwio"), dptr(bias1, "bias1"),
                                   b, h, w, dptr(ws), dptr(mean), dptr(var), stream_ptr()), "m4d_enc_level0_stats")
    return mean, var


def encoder_level0(images, w1_hwio, bias1, dn_scale, dn_bias, w2_hwio, bias2, slope=0.1, stats=None):
    """Encoder level 0 with DINL (m4depth_network.py:79-87) in one call that never writes the [b,h,w,16] map: the 3 -> 16
    convolution is recomputed from the images on the matrix cores in each pass (sums, squared deviations, normalise +
    stride-2 convolution).  16 -> 16 channels only (the reference's level 0).  ``images``: dense [b,h,w,3] or a ``FrameStack``.
    ``stats`` = (mean, var) [b,16] of these images from ``encoder_level0_stats``: only the last pass runs."""
    img_ptr, bsz, stride_b, stride_t, (b, h, w, c3), images_dev = _image_batch_address(images)
    if c3 != 3 or bias1.numel() != 16 or bias2.numel() != 16 or tuple(w2_hwio.shape) != (3, 3, 16, 16):
        raise ValueError("encoder_level0 expects RGB images and 3->16 / 16->16 convolutions")
    out = torch.empty((b, -(-h // 2), -(-w // 2), 16), dtype=torch.float32, device=images_dev)
    if stats is not None:
        mean, var = stats
        if tuple(mean.shape) != (b, 16) or tuple(var.shape) != (b, 16):
            raise ValueError(f"encoder_level0: statistics of {tuple(mean.shape)} for {b} images")
        check(lib.m4d_enc_level0_apply(img_ptr, int(bsz), int(stride_b), int(stride_t), dptr(w1_hwio, "w1_hwio"), dptr(bias1, "bias1"),
                                       dptr(mean, "mean"), dptr(var, "var"), dptr(dn_scale.reshape(-1), "dn_scale"),
                                       dptr(dn_bias.reshape(-1), "dn_bias"), float(slope), dptr(w2_hwio, "w2_hwio"),
                                       dptr(bias2, "bias2"), float(slope), b, h, w, dptr(out), stream_ptr()), "m4d_enc_level0_apply")
        return out
    ws = _workspace("dinl", 4 * int(lib.m4d_dinl_workspace_floats(b, 16)), images_dev)
    check(lib.m4d_enc_level0_fwd(img_ptr, int(bsz), int(stride_b), int(stride_t), dptr(w1_hwio, "w1_hwio"), dptr(bias1, "bias1"),
                                 dptr(dn_scale.reshape(-1), "dn_scale"), dptr(dn_bias.reshape(-1), "dn_bias"), float(slope),
                                 dptr(w2_hwio, "w2_hwio"), dptr(bias2, "bias2"), float(slope), b, h, w, dptr(ws), dptr(out),
                                 stream_ptr()), "m4d_enc_level0_fwd")
    return out


def pack_refiner_tail_weights(k6_hwio, k7_hwio):
    """Weights of the fused level tail (m4d_refiner_tail): conv6 [3,3,32,16] -> [9][16][32], conv7 [3,3,16,5] -> [9][16][16]
    (output rows 5..15 zero).  numpy in, numpy out."""
    import numpy as np
    k6 = np.asarray(k6_hwio, np.float32)
    k7 = np.asarray(k7_hwio, np.float32)
    assert k6.shape == (3, 3, 32, 16) and k7.shape == (3, 3, 16, 5), (k6.shape, k7.shape)
    w6 = np.ascontiguousarray(k6.reshape(9, 32, 16).transpose(0, 2, 1))
    w7 = np.zeros((9, 16, 16), np.float32)
    w7[:, :5, :] = k7.reshape(9, 16, 5).transpose(0, 2, 1)
    return w6, w7


def pack_refiner_tail_weights6(k6_hwio, k7_hwio):
    """Weights of the bf16-split fused level tail (m4d_refiner_tail6) as MFMA B fragments, every weight split exactly into
    three bf16 terms: conv6 [3,3,32,16] -> [9 taps][3 parts][64 lanes = (k-quarter, cout)][8 channels 8 kq .. 8 kq + 7];
    conv7 [3,3,16,5] -> [5 K-steps][3 parts][4 k-quarters][8 couts][8]: k-quarter kq = channels 8 (kq & 1) .. + 7 of tap
    2 j + (kq >> 1), zeros for the tenth tap and couts 5..7.  numpy in, numpy uint16 out."""
    import numpy as np
    k6 = np.asarray(k6_hwio, np.float32)
    k7 = np.asarray(k7_hwio, np.float32)
    assert k6.shape == (3, 3, 32, 16) and k7.shape == (3, 3, 16, 5), (k6.shape, k7.shape)
    p6 = split_bf16x3(k6.reshape(9, 4, 8, 16))                      # part, tap, kq, e, n
    w6 = np.ascontiguousarray(p6.transpose(1, 0, 2, 4, 3)).reshape(9, 3, 64, 8)        # tap, part, (kq, n), e
    full7 = np.zeros((10, 16, 8), np.float32)
    full7[:9, :, :5] = k7.reshape(9, 16, 5)
    p7 = split_bf16x3(full7.reshape(5, 2, 2, 8, 8))                 # part, j, tap parity, channel half, e, n
    w7 = np.ascontiguousarray(p7.transpose(1, 0, 2, 3, 5, 4)).reshape(5, 3, 4, 8, 8)   # j, part, kq = (parity, half), n, e
    return w6, w7


def refiner_tail6(x32, w6f, b6, w7f, b7, rot, trans, camera, scale, depth_state=None):
    """refiner_tail with float32 operands split into three bf16 terms on the bf16 matrix cores (csrc/m4d_tail6.hip);
    w6f / w7f from pack_refiner_tail_weights6 (int16 views on the device)."""
    x = as_f32(x32, "x32")
    b, h, w, c = x.shape
    if c != 32:
        raise ValueError(f"refiner_tail6 expects the 32-channel refiner activation, got {c} channels")
    rot = as_f32(rot, "rot")
    tr = as_f32(trans, "trans").reshape(b, 3)
    f = as_f32(camera["f"], "camera['f']").reshape(b, 2)
    cc = as_f32(camera["c"], "camera['c']").reshape(b, 2)
    para = torch.empty((b, h, w, 1), dtype=torch.float32, device=x.device)
    depth = torch.empty_like(para)
    other = torch.empty((b, h, w, 4), dtype=torch.float32, device=x.device)
    check(lib.m4d_refiner_tail6(dptr(x, "x32"), dptr(w6f, "w6f", torch.int16), dptr(b6, "b6"), dptr(w7f, "w7f", torch.int16),
                                dptr(b7, "b7"), dptr(rot, "rot"), rot.shape[1], dptr(tr), dptr(f), dptr(cc), b, h, w, float(scale),
                                dptr(para), dptr(depth), dptr(other), dptr(depth_state, "depth_state"), stream_ptr()),
          "m4d_refiner_tail6")
    return para, depth, other


def refiner_tail(x32, w6p, b6, w7p, b7, rot, trans, camera, scale, depth_state=None):
    """conv(32->16)+lrelu, conv(16->5) and the level tail (level_post) in one launch: returns (parallax, depth, other)."""
    x = as_f32(x32, "x32")
    b, h, w, c = x.shape
    if c != 32:
        raise ValueError(f"refiner_tail expects the 32-channel refiner activation, got {c} channels")
    rot = as_f32(rot, "rot")
    tr = as_f32(trans, "trans").reshape(b, 3)
    f = as_f32(camera["f"], "camera['f']").reshape(b, 2)
    cc = as_f32(camera["c"], "camera['c']").reshape(b, 2)
    para = torch.empty((b, h, w, 1), dtype=torch.float32, device=x.device)
    depth = torch.empty_like(para)
    other = torch.empty((b, h, w, 4), dtype=torch.float32, device=x.device)
    check(lib.m4d_refiner_tail(dptr(x, "x32"), dptr(w6p, "w6p"), dptr(b6, "b6"), dptr(w7p, "w7p"), dptr(b7, "b7"),
                               dptr(rot, "rot"), rot.shape[1], dptr(tr), dptr(f), dptr(cc), b, h, w, float(scale),
                               dptr(para), dptr(depth), dptr(other), dptr(depth_state, "depth_state"), stream_ptr()),
          "m4d_refiner_tail")
    return para, depth, other


def camera_pyramid(camera, levels):
    """Level-local intrinsics of DepthEstimatorPyramid.call (m4depth_network.py:300-302): a list (finest level first) of
    {"f": [b,2], "c": [b,2]} = camera / 2^(l+1), all levels from one launch."""
    f = as_f32(camera["f"], "camera['f']")
    c = as_f32(camera["c"], "camera['c']")
    b = f.shape[0]
    out = torch.empty((2, levels, b, 2), dtype=torch.float32, device=f.device)
    check(lib.m4d_camera_pyramid(dptr(f.reshape(b, 2), "camera['f']"), dptr(c.reshape(b, 2), "camera['c']"), b, int(levels),
                                 dptr(out[0]), dptr(out[1]), stream_ptr()), "m4d_camera_pyramid")
    return [{"f": out[0, l], "c": out[1, l]} for l in range(levels)]
