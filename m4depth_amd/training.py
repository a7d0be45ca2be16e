"""The training graph: M4Depth.call(training=True), m4depth_loss and train_step
(m4depth_network.py:278-323, 351-365, 371-431, 491-536).

Division of labour.  The reference lets tf.GradientTape differentiate its whole Python graph.
Here torch autograd records the graph, and the nodes that carry the work are HIP:

  * cost volumes: ``get_parallax_sweeping_cv`` / ``cost_volume`` forward and backward kernels
    (m4d_dscv_fwd/bwd, m4d_sncv_fwd/bwd);
  * convolutions: forward = the fp32-MFMA implicit GEMM with fused bias + leaky_relu (every layer, the 3-channel
    image layer included); data gradient = the SAME kernel run on the 180-degree-rotated, transposed
    weights (a 3x3 'SAME' stride-1 correlation is its own adjoint up to that re-packing; for a stride-2
    layer the incoming gradient is first spread onto the input grid, m4d_dilate2); weight gradient = the
    fp32-MFMA kernel of csrc/m4d_wgrad.hip (K = pixels, fixed-order two-stage reduction).  No MIOpen /
    framework convolution is left in the training step;
  * the per-pixel glue (upsampling, log/exp, parallax<->depth) is a handful of elementwise
    torch ops on the device -- they are < 1 % of a training step.

There is no CPU path: every tensor must live on the MI355X.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from . import network_ops as nops
from ._lib import lib, dptr, stream_ptr, check, as_f32
from .depth_operations import get_parallax_sweeping_cv, cost_volume, prev_d2para, depth2parallax

def _same_pads(h, w, s):
    ph = max((-(-h // s) - 1) * s + 3 - h, 0)
    pw = max((-(-w // s) - 1) * s + 3 - w, 0)
    return ph // 2, ph - ph // 2, pw // 2, pw - pw // 2


def _packed(cache, weight, transpose):
    """(wp, n_pad) of the live parameter, re-packed by one HIP kernel per optimizer step."""
    key = ("bwd" if transpose else "fwd", weight.data_ptr(), weight._version)
    hit = cache.get(key)
    if hit is None:
        O, I = weight.shape[0], weight.shape[1]
        w_ohwi = weight.detach().permute(0, 2, 3, 1).contiguous()      # a view of the channels-last parameter
        K, N = (O, I) if transpose else (I, O)
        n_pad = -(-N // 32) * 32
        wp = torch.empty((-(-K // 16), 9, n_pad, 16), dtype=torch.float32, device=weight.device)
        check(lib.m4d_pack_conv_weights(dptr(w_ohwi, "weight"), O, I, int(transpose), dptr(wp), stream_ptr()),
              "m4d_pack_conv_weights")
        hit = (wp, n_pad)
        cache[key] = hit
    return hit


def _packed_lat(cache, weight, transpose):
    """The parameter in m4d_conv3x3_lat's fragment-major bf16-split layout (int16 bits), re-packed on the device per step."""
    key = ("bwd_lat" if transpose else "fwd_lat", weight.data_ptr(), weight._version)
    hit = cache.get(key)
    if hit is None:
        O, I = weight.shape[0], weight.shape[1]
        w_ohwi = weight.detach().permute(0, 2, 3, 1).contiguous()
        K, N = (O, I) if transpose else (I, O)
        hit = torch.empty((-(-N // 32), -(-K // 16), 9, 3, 64, 8), dtype=torch.int16, device=weight.device)
        check(lib.m4d_pack_conv_weights_lat(dptr(w_ohwi, "weight"), O, I, int(transpose), dptr(hit, "wp", torch.int16), stream_ptr()),
              "m4d_pack_conv_weights_lat")
        cache[key] = hit
    return hit


def _lat_ok(b, h, w, cin, stride):
    """Small maps (the coarse levels of a training crop) run forward and data gradient on the latency-first kernel
    (csrc/m4d_convlat.hip: float32 operands as exact 3 x bf16 splits) instead of the fp32-MFMA split-K pair."""
    from . import network as net
    return (stride == 1 and net.conv_arith == "bf16x3" and cin >= 16 and cin % 4 == 0 and b * h * w <= net.lat_conv_max_pixels
            and train_lat_convs)


import os as _os
train_lat_convs = _os.environ.get("M4D_TRAIN_LAT_CONVS", "1") == "1"


class _ConvBiasAct(torch.autograd.Function):
    """leaky_relu(conv3x3_same_tf(x, w) + bias, slope) on NHWC activations, OIHW weights (channels-last strides)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, slope, cache):
        cout = weight.shape[0]
        b, h, w, cin = x.shape
        if _lat_ok(b, h, w, cin, stride):
            out = nops.conv3x3_lat(x, _packed_lat(cache, weight, False), bias.detach(), cout, slope, final=True)
        else:
            wp, n_pad = _packed(cache, weight, False)
            out = nops.conv3x3_bias_act(x, wp, bias.detach(), cout, n_pad, slope, stride=stride)
        ctx.save_for_backward(x, weight, out)
        ctx.cfg = (stride, slope, cache)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        stride, slope, cache = ctx.cfg
        b, h, w, cin = x.shape
        cout = weight.shape[0]
        oh, ow = out.shape[1:3]
        g = as_f32(g, "grad")
        rows = g.numel() // cout
        # epilogue backward: leaky_relu mask (tf.nn.leaky_relu: features > 0 ? g : alpha*g) + bias gradient, one pass
        gp = torch.empty_like(g)
        g_bias = torch.empty(cout, dtype=torch.float32, device=g.device)
        ws = nops._workspace("bias_bwd", 4 * int(lib.m4d_bias_act_bwd_workspace_floats(rows, cout)), g.device)
        check(lib.m4d_bias_act_bwd(dptr(g, "grad"), dptr(out), rows, cout, float(slope), dptr(gp), dptr(g_bias), dptr(ws),
                                   stream_ptr()), "m4d_bias_act_bwd")
        g = gp
        g_x = None
        if ctx.needs_input_grad[0]:
            # adjoint of a 'SAME' 3x3 correlation = the stride-1 correlation with k'[ky,kx,o,i] = k[2-ky,2-kx,i,o] of the
            # gradient (stride 2: of the gradient spread onto the input grid with zeros in between)
            zero = cache.get(("zero", cin))
            if zero is None:
                zero = torch.zeros(cin, dtype=torch.float32, device=x.device)
                cache[("zero", cin)] = zero
            gd = g
            if stride == 2:
                gd = torch.empty((b, h, w, cout), dtype=torch.float32, device=g.device)
                check(lib.m4d_dilate2(dptr(g), b, oh, ow, cout, h, w, dptr(gd), stream_ptr()), "m4d_dilate2")
            if _lat_ok(b, h, w, cout, 1) and stride == 1:
                g_x = nops.conv3x3_lat(gd, _packed_lat(cache, weight, True), zero, cin, 1.0, final=True)
            else:
                wp, n_pad = _packed(cache, weight, True)
                g_x = nops.conv3x3_bias_act(gd, wp, zero, cin, n_pad, 1.0, stride=1)
        g_w = None
        if ctx.needs_input_grad[1]:
            # gradient in the parameter's own memory layout: OIHW tensor with channels-last strides = [O][ky][kx][I]
            g_w = torch.empty_strided(tuple(weight.shape), (9 * cin, 1, 3 * cin, cin), dtype=torch.float32, device=x.device)
            n_ws = int(lib.m4d_conv3x3_wgrad_workspace_floats(b, h, w, cin, cout, stride))
            wsg = nops._workspace("wgrad", 4 * n_ws, x.device)
            check(lib.m4d_conv3x3_wgrad(dptr(x), dptr(g), b, h, w, cin, cout, stride, dptr(wsg), n_ws,
                                        ctypes.c_void_p(g_w.data_ptr()), stream_ptr()), "m4d_conv3x3_wgrad")
        return g_x, g_w, g_bias, None, None, None


def conv_bias_act(conv, x, slope, cache):
    """Differentiable twin of network._Conv3x3SameTF.forward."""
    if conv.weight is None:
        conv._build(x.shape[-1], x.device)
    return _ConvBiasAct.apply(as_f32(x, "x"), conv.weight, conv.bias, conv.stride, 1.0 if slope is None else slope, cache)


# ------------------------------------------------------------------------------- glue
class _UpsampleV1(torch.autograd.Function):
    """tf.compat.v1.image.resize_bilinear(x, [h, w]) * mul (legacy coordinates, m4depth_network.py:202-204)."""

    @staticmethod
    def forward(ctx, x, h, w, mul):
        ctx.cfg = (tuple(x.shape), h, w, mul)
        return nops.resize_bilinear_v1(x, h, w, mul)

    @staticmethod
    def backward(ctx, g):
        (b, ih, iw, c), h, w, mul = ctx.cfg
        g = as_f32(g, "grad")
        g_in = torch.empty((b, ih, iw, c), dtype=torch.float32, device=g.device)
        check(lib.m4d_resize_bilinear_v1_bwd(dptr(g, "grad"), b, ih, iw, c, h, w, float(mul), dptr(g_in), stream_ptr()),
              "m4d_resize_bilinear_v1_bwd")
        return g_in, None, None, None


def _upsample2_v1(x, h, w, mul=1.0):
    return _UpsampleV1.apply(as_f32(x, "x"), int(h), int(w), float(mul))


class _NormalizeCuts(torch.autograd.Function):
    """Per-cut tf.linalg.normalize (m4depth_network.py:179-189)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.save_for_backward(x)
        ctx.k = k
        return nops.normalize_cuts(x, k)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        b, h, w, c = x.shape
        g = as_f32(g, "grad")
        gx = torch.empty_like(x)
        check(lib.m4d_normalize_cuts_bwd(dptr(x), dptr(g, "grad"), b, h, w, c, ctx.k, dptr(gx), stream_ptr()),
              "m4d_normalize_cuts_bwd")
        return gx, None


def _normalize_cuts(x, k):
    return _NormalizeCuts.apply(as_f32(x, "x"), int(k))


class _LevelPost(torch.autograd.Function):
    """The level tail (m4depth_network.py:247-251): refiner output -> parallax = exp(clip)/2^m,
    depth = parallax2depth(parallax), other."""

    @staticmethod
    def forward(ctx, ro, rot, trans, f, c, scale):
        para, depth, other = nops.level_post(ro, rot, trans, {"f": f, "c": c}, scale)
        ctx.save_for_backward(ro, rot, trans, f, c)
        ctx.scale = scale
        return para, depth, other

    @staticmethod
    def backward(ctx, g_para, g_depth, g_other):
        ro, rot, trans, f, c = ctx.saved_tensors
        b, h, w, _ = ro.shape
        gs = [None if g is None else as_f32(g, "grad") for g in (g_para, g_depth, g_other)]
        g_ro = torch.empty_like(ro)
        check(lib.m4d_level_post_bwd(dptr(ro), dptr(gs[0]), dptr(gs[1]), dptr(gs[2]), dptr(rot, "rot"), rot.shape[1],
                                     dptr(trans), dptr(f), dptr(c), b, h, w, float(ctx.scale), dptr(g_ro), stream_ptr()),
              "m4d_level_post_bwd")
        return g_ro, None, None, None, None, None


class _LevelL1(torch.autograd.Function):
    """One level's unweighted term of m4depth_loss (m4depth_network.py:503-533)."""

    @staticmethod
    def forward(ctx, pred, gt, velodyne):
        b, h, w, _ = pred.shape
        H, W = gt.shape[1:3]
        ws = nops._workspace("loss", 4 * int(lib.m4d_loss_workspace_floats()), pred.device)
        out2 = torch.empty(2, dtype=torch.float32, device=pred.device)
        check(lib.m4d_loss_level_fwd(dptr(pred, "pred"), dptr(gt, "gt"), b, h, w, H, W, int(velodyne), dptr(ws), dptr(out2),
                                     stream_ptr()), "m4d_loss_level_fwd")
        ctx.save_for_backward(pred, gt, out2)
        ctx.velodyne = velodyne
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        pred, gt, out2 = ctx.saved_tensors
        b, h, w, _ = pred.shape
        H, W = gt.shape[1:3]
        g = as_f32(g, "grad").reshape(1)
        g_pred = torch.empty_like(pred)
        check(lib.m4d_loss_level_bwd(dptr(pred), dptr(gt), dptr(out2), dptr(g, "grad"), b, h, w, H, W, int(ctx.velodyne),
                                     dptr(g_pred), stream_ptr()), "m4d_loss_level_bwd")
        return g_pred, None, None


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
        para_prev_l = _upsample2_v1(prev_l_est["parallax"], h, w, 2.)
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
    rot_t = as_f32(rot, "rot")
    para_curr, depth, other = _LevelPost.apply(x, rot_t, as_f32(trans, "trans").reshape(b, 3),
                                               as_f32(camera["f"], "f").reshape(b, 2),
                                               as_f32(camera["c"], "c").reshape(b, 2), scale)   # :247-251
    return {"other": other, "depth": depth, "parallax": para_curr}


def dinl_autograd(dn, f_map):
    """DomainNormalization (m4depth_network.py:24-48) as the node the TRAINING graph records: (x - mean_hw) / (var_hw + 1e-12),
    l2-normalised over the channels, scale and bias -- torch autograd ops standing where tf.GradientTape stands (the
    inference forward runs the HIP kernel m4d_dinl_fwd_padded instead, network.DomainNormalization.forward)."""
    mean = f_map.mean(dim=(1, 2), keepdim=True)
    centred = f_map - mean
    var = (centred * centred).mean(dim=(1, 2), keepdim=True)
    n = centred / (var + 1e-12)
    ss = (n * n).sum(dim=-1, keepdim=True)
    n = n * torch.rsqrt(torch.clamp_min(ss, 1e-12))               # tf.math.l2_normalize
    return dn.scale * n + dn.bias


def encoder_forward_train(encoder, images, cache):
    """FeaturePyramid.call (m4depth_network.py:76-90), recorded by autograd."""
    fm = as_f32(images, "images")
    outs = []
    for i, (c1, c2, dn) in enumerate(zip(encoder.conv_layers_s1, encoder.conv_layers_s2, encoder.dn_layers)):
        if encoder.use_dinl and i == 0:
            t = conv_bias_act(c1, fm, None, cache)
            if dn.scale is None:
                dn._build(t.shape[-1], t.device)
            t = F.leaky_relu(dinl_autograd(dn, t), 0.1)
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


def m4depth_loss(gts, preds, depth_type="map"):
    """M4Depth.m4depth_loss (m4depth_network.py:491-536): log-depth L1 over every level of every
    frame but the first, level i weighted 0.64 / 2**(i-1); one fused HIP kernel per level."""
    l1_loss = 0.
    velodyne = depth_type == "velodyne"
    for gt, pred_pyr in zip(gts[1:], preds[1:]):
        gt_d = as_f32(gt["depth"], "gt depth")
        for i, pred in enumerate(pred_pyr):
            term = _LevelL1.apply(as_f32(pred["depth"], "pred depth"), gt_d, velodyne)
            l1_loss = l1_loss + term * ((0.64 / (2. ** (i - 1))) / float(len(gts) - 1))
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


class GraphedTrainStep:
    """One whole ``train_step`` -- forward, loss, backward, Adam update -- captured in a hipGraph.

    Eagerly a 384x384 / batch-3 / 4-frame step is several thousand kernel launches (most of them
    the few-microsecond elementwise nodes of the autograd graph) and the GPU idles two thirds of
    the time waiting for the host (profiles/r01_train_steady_state_eager.txt).  The step has no
    host synchronisation and fixed shapes, so it is captured once and replayed: the host then
    only copies the next batch into the static input buffers.

    Requirements: an optimizer whose step is capturable (``torch.optim.Adam(..., capturable=True)``),
    the batch layout of ``example`` for every later call, ``new_traj`` = first frame only (it is
    host-side control flow baked into the capture).  The eager warm-up steps (packed-weight and
    workspace allocation) are REAL optimizer steps on ``example``."""

    def __init__(self, model, example, optimizer, warmup=3, grad_sync=None):
        self.model, self.optimizer = model, optimizer
        self.static = {k: example[k].clone() for k in ("depth", "RGB_im", "rot", "trans")}
        self.camera = {k: v.clone() for k, v in example["camera"].items()}
        self.new_traj = example["new_traj"]
        self.grad_sync = grad_sync
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        from ._lib import framework_copies_in_capture
        with framework_copies_in_capture(), torch.cuda.graph(self.graph):      # the autograd graph's own nodes are framework kernels
            self.loss, self.est = self._step()

    def _data(self):
        d = dict(self.static)
        d["new_traj"] = self.new_traj
        d["camera"] = self.camera
        return d

    def _step(self):
        data = self._data()
        samples = unstack_sequence(data)
        gts = [{"depth": s["depth"]} for s in samples]
        preds = model_forward_train(self.model, samples, self.camera)
        loss = m4depth_loss(gts, preds, self.model.depth_type)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync([p for g in self.optimizer.param_groups for p in g["params"]])
        self.optimizer.step()
        return loss.detach(), preds[-1][0]["depth"].detach()

    def __call__(self, data=None):
        """Copies ``data`` into the static buffers, replays the step; returns (loss, finest depth
        estimate of the last frame) -- static device tensors, overwritten by the next replay."""
        if data is not None:
            for k in self.static:
                if data[k].data_ptr() != self.static[k].data_ptr():
                    self.static[k].copy_(data[k], non_blocking=True)
            for k in self.camera:
                if data["camera"][k].data_ptr() != self.camera[k].data_ptr():
                    self.camera[k].copy_(data["camera"][k], non_blocking=True)
        self.graph.replay()
        self.model.step_counter += 1
        return self.loss, self.est
