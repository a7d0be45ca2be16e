"""The training graph: M4Depth.call(training=True), m4depth_loss and train_step
(m4depth_network.py:278-323, 351-365, 371-431, 491-536).

Division of labour.  The reference lets tf.GradientTape differentiate its whole Python graph.
Here torch autograd records the graph, and the nodes that carry the work are HIP:

  * cost volumes: ``get_parallax_sweeping_cv`` / ``cost_volume`` forward and backward kernels
    (m4d_dscv_fwd/bwd, m4d_sncv_fwd/bwd);
  * convolutions: forward = the fp32-MFMA implicit GEMM with fused bias + leaky_relu, data
    gradient of the stride-1 layers = the SAME kernel run on the 180-degree-rotated, transposed
    weights (a 3x3 'SAME' stride-1 correlation is its own adjoint up to that re-packing);
    weight gradients and the stride-2 / 3-channel layers go through MIOpen
    (aten.convolution_backward);
  * the per-pixel glue (upsampling, log/exp, parallax<->depth) is a handful of elementwise
    torch ops on the device -- they are < 1 % of a training step.

There is no CPU path: every tensor must live on the MI355X.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import network_ops as nops
from ._lib import as_f32
from .depth_operations import get_parallax_sweeping_cv, cost_volume, prev_d2para, depth2parallax

_PERM16 = [0, 2, 4, 6, 8, 10, 12, 14, 1, 3, 5, 7, 9, 11, 13, 15]


def pack_conv_weights_device(kernel_hwio):
    """Device-side twin of network_ops.pack_conv_weights (same layout, torch ops): the weights
    change every optimizer step, so training re-packs them on the GPU instead of on the host."""
    cin, cout = kernel_hwio.shape[2], kernel_hwio.shape[3]
    nch = -(-cin // 16)
    cpad = -(-cout // 32) * 32
    full = kernel_hwio.new_zeros((9, nch * 16, cpad))
    full[:, :cin, :cout] = kernel_hwio.reshape(9, cin, cout)
    w = full.reshape(9, nch, 16, cpad)[:, :, _PERM16, :]
    return w.permute(1, 0, 3, 2).contiguous(), cpad


def _same_pads(h, w, s):
    ph = max((-(-h // s) - 1) * s + 3 - h, 0)
    pw = max((-(-w // s) - 1) * s + 3 - w, 0)
    return ph // 2, ph - ph // 2, pw // 2, pw - pw // 2


class _ConvBiasAct(torch.autograd.Function):
    """leaky_relu(conv3x3_same_tf(x, w) + bias, slope) on NHWC activations, OIHW weights."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, slope, cache):
        b, h, w, cin = x.shape
        cout = weight.shape[0]
        use_mfma = cin >= 8
        if use_mfma:
            key = ("fwd", weight.data_ptr(), weight._version)
            packed = cache.get(key)
            if packed is None:
                packed = pack_conv_weights_device(weight.detach().permute(2, 3, 1, 0))
                cache[key] = packed
            out = nops.conv3x3_bias_act(x, packed[0], bias.detach(), cout, packed[1], slope, stride=stride)
        else:
            pt, pb, pl, pr = _same_pads(h, w, stride)
            xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
            y = F.conv2d(xn, weight, None, stride, 0).permute(0, 2, 3, 1).contiguous()
            out = nops.bias_act_(y, bias.detach(), slope)
        ctx.save_for_backward(x, weight, out)
        ctx.cfg = (stride, slope, cache)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        stride, slope, cache = ctx.cfg
        b, h, w, cin = x.shape
        cout = weight.shape[0]
        g = as_f32(g, "grad")
        if slope != 1.0:
            g = g * torch.where(out > 0, 1.0, slope)          # tf.nn.leaky_relu gradient: features > 0 ? g : alpha*g
        g_bias = g.sum(dim=(0, 1, 2)) if ctx.needs_input_grad[2] else None
        need_x = ctx.needs_input_grad[0]
        g_x = None
        mfma_dgrad = need_x and stride == 1 and cout >= 8
        if mfma_dgrad:
            # adjoint of a stride-1 'SAME' 3x3 correlation = the same correlation with k'[ky,kx,o,i] = k[2-ky,2-kx,i,o]
            key = ("bwd", weight.data_ptr(), weight._version)
            packed = cache.get(key)
            if packed is None:
                packed = pack_conv_weights_device(weight.detach().flip(2, 3).permute(2, 3, 0, 1))
                cache[key] = packed
            zero = cache.get(("zero", cin))
            if zero is None:
                zero = torch.zeros(cin, dtype=torch.float32, device=x.device)
                cache[("zero", cin)] = zero
            g_x = nops.conv3x3_bias_act(g, packed[0], zero, cin, packed[1], 1.0, stride=1)
        pt, pb, pl, pr = _same_pads(h, w, stride)
        xn = x.permute(0, 3, 1, 2)
        if pt or pb or pl or pr:
            xn = F.pad(xn, (pl, pr, pt, pb))
        mask = [bool(need_x and not mfma_dgrad), bool(ctx.needs_input_grad[1]), False]
        g_w = None
        if mask[0] or mask[1]:
            gi, g_w, _ = torch.ops.aten.convolution_backward(
                g.permute(0, 3, 1, 2), xn, weight, None, [stride, stride], [0, 0], [1, 1], False, [0, 0], 1, mask)
            if mask[0]:
                g_x = gi[:, :, pt:pt + h, pl:pl + w].permute(0, 2, 3, 1).contiguous()
        return g_x, g_w, g_bias, None, None, None


def conv_bias_act(conv, x, slope, cache):
    """Differentiable twin of network._Conv3x3SameTF.forward."""
    if conv.weight is None:
        conv._build(x.shape[-1], x.device)
    return _ConvBiasAct.apply(as_f32(x, "x"), conv.weight, conv.bias, conv.stride, 1.0 if slope is None else slope, cache)


# ------------------------------------------------------------------------------- glue
def _upsample2_v1(x, h, w):
    """tf.compat.v1.image.resize_bilinear(x, [h, w]) (legacy coordinates, m4depth_network.py:202-204)
    with differentiable torch ops: src = dst * in/out, lower = floor, upper = min(ceil, in-1)."""
    b, ih, iw, c = x.shape

    def weights(out_n, in_n):
        src = torch.arange(out_n, dtype=torch.float32, device=x.device) * (np.float32(in_n) / np.float32(out_n))
        fl = torch.floor(src)
        lo = fl.to(torch.int64).clamp_(0, in_n - 1)
        hi = torch.ceil(src).to(torch.int64).clamp_(0, in_n - 1)
        return lo, hi, src - fl

    ylo, yhi, yl = weights(h, ih)
    xlo, xhi, xl = weights(w, iw)
    xl = xl.reshape(1, 1, w, 1)
    yl = yl.reshape(1, h, 1, 1)
    top_rows, bot_rows = x[:, ylo], x[:, yhi]
    tl, tr = top_rows[:, :, xlo], top_rows[:, :, xhi]
    bl, br = bot_rows[:, :, xlo], bot_rows[:, :, xhi]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return top + (bot - top) * yl


def _normalize_cuts(x, k):
    b, h, w, c = x.shape
    xr = x.reshape(b, h, w, k, c // k)
    return (xr / torch.sqrt((xr * xr).sum(dim=-1, keepdim=True))).reshape(b, h, w, c)


def _parallax_factors(b, h, w, rot, trans, camera):
    """(s, tz, alpha) of parallax2depth (utils/depth_operations.py:146-162) as [b,h,w,1] maps, in the
    operand order of the HIP converter (m4d_common.h: m4d_pixel_factors).  They depend on the camera
    motion only, not on the network, so nothing here records a gradient."""
    dev = trans.device
    f = as_f32(camera["f"], "camera['f']").reshape(b, 2)
    c = as_f32(camera["c"], "camera['c']").reshape(b, 2)
    t = as_f32(trans, "trans").reshape(b, 3)
    from .depth_operations import get_rot_mat
    R = get_rot_mat(as_f32(rot, "rot"))
    fx, fy = f[:, 0].reshape(b, 1, 1), f[:, 1].reshape(b, 1, 1)
    x = ((torch.arange(w, dtype=torch.float32, device=dev) + 0.5).reshape(1, 1, w) - c[:, 0].reshape(b, 1, 1)) / fx
    y = ((torch.arange(h, dtype=torch.float32, device=dev) + 0.5).reshape(1, h, 1) - c[:, 1].reshape(b, 1, 1)) / fy

    def row(k):
        return (R[:, k, 0].reshape(b, 1, 1) * x + R[:, k, 1].reshape(b, 1, 1) * y) + R[:, k, 2].reshape(b, 1, 1)

    rcx, rcy, alpha = row(0), row(1), row(2)
    proj_x = (rcx * fx) / alpha
    proj_y = (rcy * fy) / alpha
    tz = t[:, 2].reshape(b, 1, 1)
    dx = (t[:, 0] * f[:, 0]).reshape(b, 1, 1) - tz * proj_x
    dy = (t[:, 1] * f[:, 1]).reshape(b, 1, 1) - tz * proj_y
    s = torch.sqrt(dx * dx + dy * dy)
    return s.unsqueeze(-1), tz.unsqueeze(-1), alpha.unsqueeze(-1)


def level_forward_train(level, curr_f_maps, prev_l_est, rot, trans, camera, prev_f_maps, prev_t_depth, cache):
    """DepthEstimatorLevel.call with is_training=True (m4depth_network.py:167-262)."""
    b, h, w, c = curr_f_maps.shape
    dev = curr_f_maps.device
    k = level.nbre_cuts
    vp = (lambda t: _normalize_cuts(t, k)) if level.ablation.normalize_features else (lambda t: t)
    curr_f = vp(curr_f_maps)
    if prev_l_est is None:                                                           # :196-200
        para_prev_l = torch.ones((b, h, w, 1), device=dev)
        depth_prev_l = torch.full((b, h, w, 1), 1000., device=dev)
        other_prev_l = torch.zeros((b, h, w, 4), device=dev)
    else:                                                                            # :202-204
        other_prev_l = _upsample2_v1(prev_l_est["other"], h, w)
        para_prev_l = _upsample2_v1(prev_l_est["parallax"], h, w) * 2.
        depth_prev_l = _upsample2_v1(prev_l_est["depth"], h, w)
    if prev_t_depth is None:                                                         # :208-214
        return {"depth": depth_prev_l, "parallax": para_prev_l, "other": other_prev_l}
    prev_f = vp(prev_f_maps)
    scale = float(2.0 ** level.lvl_mul)
    with torch.no_grad():                                                            # tf.stop_gradient, depth_operations.py:215
        para_prev_t = prev_d2para(prev_t_depth.detach().contiguous(), rot, trans, camera)
    cv, para_reproj = get_parallax_sweeping_cv(curr_f.contiguous(), prev_f.contiguous(), para_prev_t,
                                               para_prev_l.contiguous(), rot, trans, camera, level.dscv_range,
                                               nbre_cuts=k, cv_accum=level.cv_accum)
    feats = [cv, torch.log(para_prev_l * scale)]                                      # :224
    if level.ablation.level_memory:
        feats.append(other_prev_l)
    if level.ablation.SNCV:
        cf = curr_f.contiguous()
        feats.append(cost_volume(cf, cf, level.sncv_range, nbre_cuts=k))             # :232
    if level.ablation.time_recurr:
        r = level.dscv_range
        feats.append(torch.log(para_reproj[..., r:r + 1] * scale))                   # :238
    x = torch.cat(feats, dim=3)
    convs = list(level.disp_refiner.prep_conv_layers) + list(level.disp_refiner.est_d_conv_layers)
    for i, conv in enumerate(convs):
        x = conv_bias_act(conv, x, 0.1 if i < len(convs) - 1 else None, cache)
    para, other = x[..., :1], x[..., 1:]
    para_curr = torch.exp(torch.clamp(para, -7., 7.)) / scale                        # :250
    s, tz, alpha = _parallax_factors(b, h, w, rot, trans, camera)
    depth = (s / para_curr - tz) / alpha                                             # parallax2depth, :251
    return {"other": other, "depth": depth, "parallax": para_curr}


def encoder_forward_train(encoder, images, cache):
    """FeaturePyramid.call (m4depth_network.py:76-90), recorded by autograd."""
    fm = as_f32(images, "images")
    outs = []
    for i, (c1, c2, dn) in enumerate(zip(encoder.conv_layers_s1, encoder.conv_layers_s2, encoder.dn_layers)):
        if encoder.use_dinl and i == 0:
            t = conv_bias_act(c1, fm, None, cache)
            if dn.scale is None:
                dn._build(t.shape[-1], t.device)
            t = F.leaky_relu(dn._forward_torch(t), 0.1)
        else:
            t = conv_bias_act(c1, fm, 0.1, cache)
        fm = conv_bias_act(c2, t, 0.1, cache)
        outs.append(fm)
    return outs


def model_forward_train(model, traj_samples, camera):
    """M4Depth.call(training=True): returns d_est_seq[seq][level, fine -> coarse] (m4depth_network.py:351-365)."""
    cache = model.__dict__.setdefault("_train_cache", {})
    for key in [k for k in cache if k[0] in ("fwd", "bwd")]:
        del cache[key]                                 # weights may have been stepped since the last call
    pyrs = [encoder_forward_train(model.encoder, s["RGB_im"], cache) for s in traj_samples]
    levels = model.d_estimator.levels
    L = len(levels)
    cams = [{"f": as_f32(camera["f"], "f") / 2. ** (lvl + 1), "c": as_f32(camera["c"], "c") / 2. ** (lvl + 1)}
            for lvl in range(L)]
    d_est_seq = []
    for seq_i, (f_pyr, sample) in enumerate(zip(pyrs, traj_samples)):
        d_est_curr = None
        for l in range(L):
            lvl = L - 1 - l
            f_prev = d_prev = None
            if seq_i != 0:                                                           # :297-299
                f_prev = pyrs[seq_i - 1][lvl]
                d_prev = d_est_seq[-1][lvl]["depth"]
            prev = None if d_est_curr is None else dict(d_est_curr[-1])
            est = level_forward_train(levels[lvl], f_pyr[lvl], prev, sample["rot"], sample["trans"], cams[lvl],
                                      f_prev, d_prev, cache)
            d_est_curr = [est] if d_est_curr is None else d_est_curr + [est]
        d_est_seq.append(d_est_curr[::-1])
    return d_est_seq


def _resize_half_pixel(x, h, w):
    """tf.image.resize(x, [h, w]) (bilinear, half-pixel centres, no antialias; m4depth_network.py:532)."""
    b, ih, iw, c = x.shape

    def weights(out_n, in_n):
        src = (torch.arange(out_n, dtype=torch.float32, device=x.device) + 0.5) * (in_n / out_n) - 0.5
        fl = torch.floor(src)
        return fl.to(torch.int64).clamp_(0, in_n - 1), torch.ceil(src).to(torch.int64).clamp_(0, in_n - 1), src - fl

    ylo, yhi, yl = weights(h, ih)
    xlo, xhi, xl = weights(w, iw)
    xl = xl.reshape(1, 1, w, 1)
    yl = yl.reshape(1, h, 1, 1)
    tl, tr = x[:, ylo][:, :, xlo], x[:, ylo][:, :, xhi]
    bl, br = x[:, yhi][:, :, xlo], x[:, yhi][:, :, xhi]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return top + (bot - top) * yl


def m4depth_loss(gts, preds, depth_type="map"):
    """M4Depth.m4depth_loss (m4depth_network.py:491-536): log-depth L1 over every level of every
    frame but the first, level i weighted 0.64 / 2**(i-1)."""
    def preprocess(x):
        return torch.log(torch.clamp(x, 0.01, 200.))

    def masked_reduce_mean(array, mask, dims=None):
        if dims is None:
            return (array * mask).sum() / (mask.sum() + 1e-12)
        return (array * mask).sum(dim=dims) / (mask.sum(dim=dims) + 1e-12)

    l1_loss = 0.
    for gt, pred_pyr in zip(gts[1:], preds[1:]):
        gt_pre = preprocess(gt["depth"])
        for i, pred in enumerate(pred_pyr):
            pred_depth = preprocess(pred["depth"])
            b, h, w = pred_depth.shape[:3]
            if depth_type == "velodyne":
                h_g, w_g = gt_pre.shape[1:3]
                mask = (gt["depth"].reshape(b, h, h_g // h, w, w_g // w, 1) > 0).float()
                gt_resized = masked_reduce_mean(gt_pre.reshape(b, h, h_g // h, w, w_g // w, 1), mask, dims=(2, 4))
                new_mask = (mask.sum(dim=(2, 4)) > 0.).float()
                term = (0.64 / (2. ** (i - 1))) * masked_reduce_mean(torch.abs(gt_resized - pred_depth), new_mask)
            else:
                gt_resized = _resize_half_pixel(gt_pre, h, w)
                term = (0.64 / (2. ** (i - 1))) * torch.abs(gt_resized - pred_depth).mean()
            l1_loss = l1_loss + term / float(len(gts) - 1)
    return l1_loss


def unstack_sequence(data):
    """The sample re-arrangement at the top of train_step / test_step (m4depth_network.py:377-385)."""
    T = data["depth"].shape[1]
    out = [{} for _ in range(T)]
    for key in ("depth", "RGB_im", "new_traj", "rot", "trans"):
        for i in range(T):
            out[i][key] = data[key][:, i]
    return out


def set_trainable(model, flag=True):
    """Keras marks every conv kernel / bias and the DINL scale / bias trainable; the inference
    build keeps them frozen (requires_grad=False) so that no autograd graph is ever recorded."""
    for p in model.parameters():
        p.requires_grad_(flag)
    return model


def train_step(model, data, optimizer, grad_sync=None):
    """M4Depth.train_step (m4depth_network.py:371-431): loss, gradients, optimizer update, metrics
    of the last frame.  ``data`` holds [b,T,...] device tensors (+ data['camera']).  ``grad_sync``
    (optional) runs between backward and the update -- dist.all_reduce_gradients for data-parallel
    training over RCCL."""
    from .metrics import RootMeanSquaredLogError
    traj_samples = unstack_sequence(data)
    with torch.no_grad():
        gts = [{"depth": s["depth"], "parallax": depth2parallax(s["depth"], s["rot"], s["trans"], data["camera"])}
               for s in traj_samples]
    with torch.enable_grad():
        preds = model_forward_train(model, traj_samples, data["camera"])
        loss = m4depth_loss(gts, preds, model.depth_type)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if grad_sync is not None:
        grad_sync([p for g in optimizer.param_groups for p in g["params"]])
    optimizer.step()
    model.step_counter += 1
    with torch.no_grad():                                                            # :418-427
        gt = gts[-1]["depth"]
        est = nops.resize_nearest(preds[-1][0]["depth"].detach().contiguous(), gt.shape[1], gt.shape[2])
        gt_c = torch.clamp(gt, 0.0, 80.)
        est_c = torch.clamp(est, 0.001, 80.)
        if not getattr(model, "compiled_metrics", None):
            model.compiled_metrics = [RootMeanSquaredLogError()]
        for m in model.compiled_metrics:
            m.update_state(gt_c, est_c)
    out = {m.name: m.result() for m in model.compiled_metrics}
    out["loss"] = loss.detach()
    return out
