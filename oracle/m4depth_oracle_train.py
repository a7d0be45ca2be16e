"""CPU ORACLE for the TRAINING graph — test infrastructure only, never a product path.

The reference differentiates its Python/TensorFlow graph with ``tf.GradientTape``
(train_step, m4depth_network.py:371-399).  TensorFlow cannot run here (PARITY UNPINNED,
see m4depth_oracle.py), so this module restates the same graph op for op with
torch **CPU** tensors and lets torch's autodiff play the role of TF's: gathers become
scatter-adds, ``floor`` has no gradient, ``clip_by_value`` passes inside its bounds, the
float16 product / mean of the DSCV (utils/depth_operations.py:276-277) are differentiated in
float16, ``prev_d2para`` ends in ``stop_gradient`` (:215).  The forward values are checked
against the strict-float32 numpy oracle in tests/test_oracle_train.py.

Only ``tests/`` and ``__graft_entry__.smoke()`` may import this module.
Conventions as in m4depth_oracle.py: NHWC, ``i`` = column (x), ``j`` = row (y); conv kernels
are TF HWIO tensors (the trainable variables of the reference).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

ENCODER_CHANNELS = [16, 32, 64, 96, 128, 192]          # m4depth_network.py:59
REFINER_CHANNELS = [128, 128, 96, 64, 32, 16, 5]       # m4depth_network.py:103,109


def _t(x, dtype=torch.float32):
    return x.to(dtype) if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=dtype)


# --------------------------------------------------------------------------
# geometry (utils/depth_operations.py)
# --------------------------------------------------------------------------
def get_rot_mat(rot):
    """utils/depth_operations.py:18-53."""
    b, c = rot.shape
    if c == 3:
        one = torch.ones(b, dtype=rot.dtype)
        m = torch.stack((one, -rot[:, 2], rot[:, 1], rot[:, 2], one, -rot[:, 0], -rot[:, 1], rot[:, 0], one), -1)
        return m.reshape(b, 3, 3)
    if c != 4:
        raise ValueError("Rotation must be expressed as a small angle (x,y,z) or a quaternion (w,x,y,z)")
    w, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    m = torch.stack((1.0 - (tyy + tzz), txy - twz, txz + twy,
                     txy + twz, 1.0 - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, 1.0 - (txx + tyy)), -1)
    return m.reshape(b, 3, 3)


def motion_factors(b, h, w, rot, trans, camera, dtype=torch.float32):
    """The block shared by parallax2depth (:146-162), depth2parallax (:174-190) and the DSCV
    (:239-261); same operand order as m4depth_oracle.motion_factors."""
    f = _t(camera["f"], dtype).reshape(b, 2)
    c = _t(camera["c"], dtype).reshape(b, 2)
    t = _t(trans, dtype).reshape(b, 3)
    R = get_rot_mat(_t(rot, dtype))
    gx = (torch.arange(w, dtype=dtype) + 0.5).reshape(1, 1, w) - c[:, 0].reshape(b, 1, 1)
    gy = (torch.arange(h, dtype=dtype) + 0.5).reshape(1, h, 1) - c[:, 1].reshape(b, 1, 1)
    fx, fy = f[:, 0].reshape(b, 1, 1), f[:, 1].reshape(b, 1, 1)
    x = (gx / fx).expand(b, h, w)
    y = (gy / fy).expand(b, h, w)

    def row(k):
        return (R[:, k, 0].reshape(b, 1, 1) * x + R[:, k, 1].reshape(b, 1, 1) * y) + R[:, k, 2].reshape(b, 1, 1)

    rcx, rcy, rcz = row(0), row(1), row(2)
    alpha = rcz
    proj_x = (rcx * fx) / alpha
    proj_y = (rcy * fy) / alpha
    stx = (t[:, 0] * f[:, 0]).reshape(b, 1, 1)
    sty = (t[:, 1] * f[:, 1]).reshape(b, 1, 1)
    stz = t[:, 2].reshape(b, 1, 1)
    delta_x = stx - stz * proj_x
    delta_y = sty - stz * proj_y
    s = torch.sqrt(delta_x * delta_x + delta_y * delta_y)
    return dict(alpha=alpha, proj_x=proj_x, proj_y=proj_y, delta_x=delta_x, delta_y=delta_y, sqrt=s,
                start_x=x * fx, start_y=y * fy, stz=stz, x=x, y=y, fx=fx, fy=fy, stx=stx, sty=sty)


def parallax2depth(disp, rot, trans, camera):
    """utils/depth_operations.py:141-166."""
    b, h, w = disp.shape[:3]
    m = motion_factors(b, h, w, rot, trans, camera, disp.dtype)
    return ((m["sqrt"] / disp[..., 0] - m["stz"]) / m["alpha"]).unsqueeze(-1)


def depth2parallax(depth, rot, trans, camera):
    """utils/depth_operations.py:169-194."""
    b, h, w = depth.shape[:3]
    m = motion_factors(b, h, w, rot, trans, camera, depth.dtype)
    return (m["sqrt"] / (depth[..., 0] * m["alpha"] + m["stz"])).unsqueeze(-1)


def prev_d2para(prev_d, rot, trans, camera):
    """utils/depth_operations.py:197-215; the result is wrapped in tf.stop_gradient (:215)."""
    b, h, w = prev_d.shape[:3]
    m = motion_factors(b, h, w, rot, trans, camera, prev_d.dtype)
    cx, cy = m["x"] * m["fx"], m["y"] * m["fy"]
    den = prev_d[..., 0] - m["stz"]
    dx = (m["stx"] - m["stz"] * cx) / den
    dy = (m["sty"] - m["stz"] * cy) / den
    return torch.sqrt(dx * dx + dy * dy).unsqueeze(-1).detach()


# --------------------------------------------------------------------------
# utils/dense_image_warp.py
# --------------------------------------------------------------------------
def interpolate_bilinear(grid, query):
    """_interpolate_bilinear, utils/dense_image_warp.py:61-192, indexing 'ij'.
    grid [B,H,W,C], query [B,N,2] (row, col) -> [B,N,C]."""
    B, H, W, C = grid.shape
    alphas, floors, ceils = [], [], []
    for dim, size in ((0, H), (1, W)):
        q = query[..., dim]
        fl = torch.minimum(torch.maximum(torch.zeros((), dtype=q.dtype), torch.floor(q)),
                           torch.tensor(float(size - 2), dtype=q.dtype))         # :138-141 (floor: no gradient)
        fl = fl.detach()
        i0 = fl.to(torch.int64)
        floors.append(i0)
        ceils.append(i0 + 1)
        alphas.append(torch.clamp(q - fl, 0.0, 1.0).unsqueeze(-1))              # :147-154
    flat = grid.reshape(B, H * W, C)

    def gather(yc, xc):
        lin = (yc * W + xc).unsqueeze(-1).expand(B, -1, C)                       # :165-178
        return torch.gather(flat, 1, lin)

    tl = gather(floors[0], floors[1])
    tr = gather(floors[0], ceils[1])
    bl = gather(ceils[0], floors[1])
    br = gather(ceils[0], ceils[1])
    top = alphas[1] * (tr - tl) + tl                                            # :188-190
    bot = alphas[1] * (br - bl) + bl
    return alphas[0] * (bot - top) + top


def get_parallax_sweeping_cv(c1, c2, disp_prev_t, disp, rot, trans, camera, search_range, nbre_cuts=1,
                             half=True):
    """DSCV, utils/depth_operations.py:224-281.  ``half=False`` keeps the correlation in the
    working dtype (used for float64 gradient checks)."""
    b, h, w, C = c1.shape
    r = int(search_range)
    ncp = 2 * r + 1
    k = int(nbre_cuts)
    n_c = C // k
    dt = c1.dtype
    m = motion_factors(b, h, w, rot, trans, camera, dt)
    src = torch.cat([c2, disp_prev_t], dim=-1)                                   # :268
    jj = torch.arange(h, dtype=dt).reshape(1, h, 1).expand(b, h, w)
    ii = torch.arange(w, dtype=dt).reshape(1, 1, w).expand(b, h, w)
    cvs = [[None] * ncp for _ in range(k)]
    prev = []
    for t in range(ncp):
        p = torch.clamp(disp[..., 0] + float(t - r), 1e-6, 1e6)                  # :235-236
        divider = m["sqrt"] / p                                                  # :262
        dxx = m["delta_x"] / divider
        dyy = m["delta_y"] / divider
        flow_x = (m["proj_x"] + dxx) - m["start_x"]                              # :264
        flow_y = (m["proj_y"] + dyy) - m["start_y"]
        query = torch.stack([jj + flow_y, ii + flow_x], dim=-1).reshape(b, h * w, 2)
        wv = interpolate_bilinear(src, query).reshape(b, h, w, C + 1)
        prev.append(wv[..., C])
        if half:
            prod = c1.to(torch.float16) * wv[..., :C].to(torch.float16)         # :276
        else:
            prod = c1 * wv[..., :C]
        for kk in range(k):
            val = prod[..., kk * n_c:(kk + 1) * n_c].mean(dim=-1)                # :277
            cvs[kk][t] = val.to(dt)
    cv = torch.stack([cvs[kk][t] for kk in range(k) for t in range(ncp)], dim=-1)    # cut-major (:278)
    return cv, torch.stack(prev, dim=-1)


def cost_volume(c1, c2, search_range, dilation_rate=1, nbre_cuts=1):
    """SNCV, utils/depth_operations.py:284-313."""
    b, h, w, C = c1.shape
    r = int(search_range) * dilation_rate
    mo = 2 * int(search_range) + 1
    k = int(nbre_cuts)
    n_c = C // k
    padded = F.pad(c2, (0, 0, r, r, r, r))                                       # :293 (W then H from the back)
    outs = []
    for y in range(mo):
        for x in range(mo):
            sl = padded[:, y * dilation_rate:y * dilation_rate + h, x * dilation_rate:x * dilation_rate + w, :]
            prod = c1 * sl
            for kk in range(k):
                outs.append(prod[..., kk * n_c:(kk + 1) * n_c].mean(dim=-1))
    return F.leaky_relu(torch.stack(outs, dim=-1), 0.1)


# --------------------------------------------------------------------------
# m4depth_network.py
# --------------------------------------------------------------------------
def normalize_cuts(x, nbre_cuts):
    """m4depth_network.py:179-189 (tf.linalg.normalize: x / sqrt(sum x^2), no epsilon)."""
    b, h, w, C = x.shape
    xr = x.reshape(b, h, w, nbre_cuts, C // nbre_cuts)
    return (xr / torch.sqrt((xr * xr).sum(dim=-1, keepdim=True))).reshape(b, h, w, C)


def _resize_weights(out_n, in_n, half_pixel, dtype):
    scale = in_n / out_n
    dst = torch.arange(out_n, dtype=dtype)
    src = (dst + 0.5) * scale - 0.5 if half_pixel else dst * scale
    fl = torch.floor(src)
    lo = torch.clamp(fl.to(torch.int64), 0, in_n - 1)
    hi = torch.clamp(torch.ceil(src).to(torch.int64), 0, in_n - 1)
    return lo, hi, src - fl


def resize_bilinear(x, out_h, out_w, half_pixel):
    """half_pixel=False: tf.compat.v1.image.resize_bilinear defaults (m4depth_network.py:202-204);
    half_pixel=True: tf.image.resize(bilinear, antialias=False) as in m4depth_loss (:532)."""
    b, ih, iw, c = x.shape
    ylo, yhi, yl = _resize_weights(out_h, ih, half_pixel, x.dtype)
    xlo, xhi, xl = _resize_weights(out_w, iw, half_pixel, x.dtype)
    xl = xl.reshape(1, 1, out_w, 1)
    yl = yl.reshape(1, out_h, 1, 1)
    tl = x[:, ylo][:, :, xlo]
    tr = x[:, ylo][:, :, xhi]
    bl = x[:, yhi][:, :, xlo]
    br = x[:, yhi][:, :, xhi]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return top + (bot - top) * yl


def conv2d_same(x, kernel_hwio, bias, stride):
    """Keras Conv2D(padding='same') on NHWC (m4depth_network.py:63-72,104-114): TF 'SAME'
    puts the odd padding pixel at the bottom / right."""
    b, h, w, _ = x.shape
    kh, kw = kernel_hwio.shape[:2]
    oh, ow = -(-h // stride), -(-w // stride)
    ph = max((oh - 1) * stride + kh - h, 0)
    pw = max((ow - 1) * stride + kw - w, 0)
    xn = F.pad(x.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    y = F.conv2d(xn, kernel_hwio.permute(3, 2, 0, 1), bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def domain_normalization(x, scale, bias):
    """m4depth_network.py:44-48 ((x-mean)/(var+1e-12), then tf.math.l2_normalize over channels)."""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) * (x - mean)).mean(dim=(1, 2), keepdim=True)
    n = (x - mean) / (var + 1e-12)
    ss = (n * n).sum(dim=-1, keepdim=True)
    n = n * torch.rsqrt(torch.clamp(ss, min=1e-12))
    return scale.reshape(1, 1, 1, -1) * n + bias.reshape(1, 1, 1, -1)


def feature_pyramid(images, weights, nbre_lvls, use_dinl=True):
    """FeaturePyramid.call, m4depth_network.py:76-90."""
    fm = images
    outs = []
    for i in range(nbre_lvls):
        t = conv2d_same(fm, weights[f"enc.s1.{i}.kernel"], weights[f"enc.s1.{i}.bias"], 1)
        if use_dinl and i == 0:
            t = domain_normalization(t, weights["enc.dn.0.scale"], weights["enc.dn.0.bias"])
        t = F.leaky_relu(t, 0.1)
        t = conv2d_same(t, weights[f"enc.s2.{i}.kernel"], weights[f"enc.s2.{i}.bias"], 2)
        fm = F.leaky_relu(t, 0.1)
        outs.append(fm)
    return outs


def disp_refiner(f_input, weights, lvl):
    """DispRefiner.call, m4depth_network.py:116-135."""
    x = f_input
    n = len(REFINER_CHANNELS)
    for i in range(n):
        x = conv2d_same(x, weights[f"lvl.{lvl}.conv.{i}.kernel"], weights[f"lvl.{lvl}.conv.{i}.bias"], 1)
        if i < n - 1:
            x = F.leaky_relu(x, 0.1)
    return x


DEFAULT_ABLATION = dict(DINL=True, SNCV=True, time_recurr=True, normalize_features=True,
                        subdivide_features=True, level_memory=True)


def level_train(weights, lvl_depth, curr_f_maps, prev_l_est, rot, trans, camera, prev_f_maps, prev_t_depth,
                ablation, dscv_range=4, sncv_range=3, half=True):
    """DepthEstimatorLevel.call with is_training=True (m4depth_network.py:167-262): the previous
    frame's features and depth estimate arrive as arguments instead of state variables."""
    b, h, w, c = curr_f_maps.shape
    dt = curr_f_maps.dtype
    k = 2 ** (lvl_depth // 2) if ablation["subdivide_features"] else 1
    vp = (lambda t: normalize_cuts(t, k)) if ablation["normalize_features"] else (lambda t: t)
    curr_f = vp(curr_f_maps)
    if prev_f_maps is not None:
        prev_f = vp(prev_f_maps)
    if prev_l_est is None:
        para_prev_l = torch.ones(b, h, w, 1, dtype=dt)
        depth_prev_l = 1000. * torch.ones(b, h, w, 1, dtype=dt)
        other_prev_l = torch.zeros(b, h, w, 4, dtype=dt)
    else:
        other_prev_l = resize_bilinear(prev_l_est["other"], h, w, False)
        para_prev_l = resize_bilinear(prev_l_est["parallax"], h, w, False) * 2.
        depth_prev_l = resize_bilinear(prev_l_est["depth"], h, w, False)
    if prev_t_depth is None:                                                     # :208-214
        return {"depth": depth_prev_l, "parallax": para_prev_l, "other": other_prev_l}
    lvl_mul = lvl_depth - 3
    scale = 2.0 ** lvl_mul
    para_prev_t = prev_d2para(prev_t_depth, rot, trans, camera)                  # :218
    cv, para_reproj = get_parallax_sweeping_cv(curr_f, prev_f, para_prev_t, para_prev_l, rot, trans, camera,
                                               dscv_range, nbre_cuts=k, half=half)
    feats = [cv, torch.log(para_prev_l * scale)]
    if ablation["level_memory"]:
        feats.append(other_prev_l)
    if ablation["SNCV"]:
        feats.append(cost_volume(curr_f, curr_f, sncv_range, nbre_cuts=k))
    if ablation["time_recurr"]:
        feats.append(torch.log(para_reproj[..., dscv_range:dscv_range + 1] * scale))
    f_input = torch.cat(feats, dim=3)
    out = disp_refiner(f_input, weights, lvl_depth)
    para, other = out[..., :1], out[..., 1:]
    para_curr = torch.exp(torch.clamp(para, -7., 7.)) / scale                    # :250
    depth = parallax2depth(para_curr, rot, trans, camera)
    return {"other": other, "depth": depth, "parallax": para_curr}


def model_train(weights, traj_samples, camera, nbre_levels, ablation=None, dscv_range=4, sncv_range=3,
                half=True):
    """M4Depth.call(training=True) + DepthEstimatorPyramid.call (m4depth_network.py:278-323,351-365):
    returns d_est_seq[seq][level fine->coarse]."""
    ablation = dict(DEFAULT_ABLATION, **(ablation or {}))
    L = nbre_levels
    pyrs = [feature_pyramid(s["RGB_im"], weights, L, ablation["DINL"]) for s in traj_samples]
    d_est_seq = []
    for seq_i, (f_pyr, sample) in enumerate(zip(pyrs, traj_samples)):
        cnter = float(L)
        d_est_curr = None
        for l in range(L):
            lvl = L - 1 - l
            f_prev = d_prev = None
            if seq_i != 0:                                                       # :297-299
                f_prev = pyrs[seq_i - 1][lvl]
                d_prev = d_est_seq[-1][lvl]["depth"]
            cam = {"f": _t(camera["f"], f_pyr[lvl].dtype) / 2. ** cnter,
                   "c": _t(camera["c"], f_pyr[lvl].dtype) / 2. ** cnter}
            prev = None if d_est_curr is None else dict(d_est_curr[-1])
            est = level_train(weights, lvl + 1, f_pyr[lvl], prev, sample["rot"], sample["trans"], cam, f_prev,
                              d_prev, ablation, dscv_range, sncv_range, half)
            d_est_curr = [est] if d_est_curr is None else d_est_curr + [est]
            cnter -= 1.
        d_est_seq.append(d_est_curr[::-1])
    return d_est_seq


def m4depth_loss(gts, preds, depth_type="map"):
    """M4Depth.m4depth_loss, m4depth_network.py:491-536."""
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
                tmp = gt["depth"].reshape(b, h, h_g // h, w, w_g // w, 1)
                mask = (tmp > 0).to(pred_depth.dtype)
                tmp = gt_pre.reshape(b, h, h_g // h, w, w_g // w, 1)
                gt_resized = masked_reduce_mean(tmp, mask, dims=(2, 4))
                new_mask = (mask.sum(dim=(2, 4)) > 0.).to(pred_depth.dtype)
                term = (0.64 / (2. ** (i - 1))) * masked_reduce_mean(torch.abs(gt_resized - pred_depth), new_mask)
            else:
                gt_resized = resize_bilinear(gt_pre, h, w, True)
                term = (0.64 / (2. ** (i - 1))) * torch.abs(gt_resized - pred_depth).mean()
            l1_loss = l1_loss + term / float(len(gts) - 1)
    return l1_loss


def train_loss(weights, data, nbre_levels, ablation=None, dscv_range=4, sncv_range=3, depth_type="map",
               half=True):
    """The taped part of train_step (m4depth_network.py:374-391): data holds [b,T,...] tensors."""
    T = data["depth"].shape[1]
    samples = [{k: data[k][:, i] for k in ("depth", "RGB_im", "new_traj", "rot", "trans")} for i in range(T)]
    gts = [{"depth": s["depth"],
            "parallax": depth2parallax(s["depth"], s["rot"], s["trans"], data["camera"])} for s in samples]
    preds = model_train(weights, samples, data["camera"], nbre_levels, ablation, dscv_range, sncv_range, half)
    return m4depth_loss(gts, preds, depth_type), preds
