"""CPU ORACLE — test infrastructure only, never a product path.

A strict-IEEE float32 numpy restatement of M4Depth's per-frame parallax
cost-volume inference path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product
(``m4depth_amd``) never does and fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference (michael-fonder/M4Depth) is Python on
TensorFlow 2.7; TensorFlow is neither installed nor installable here, the
reference ships no tests / golden vectors / weights, and its only native code
(cuda_backproject) needs nvcc + TF headers.  This file therefore restates the
algorithm from the reference sources (each function cites the file:line it
follows, relative to the reference root) and from TF's documented op semantics;
points where TF's internal arithmetic order is not derivable from the
reference are marked [UNPINNED] and the choice made here is stated.

Conventions: NHWC float32; ``i`` = column (x), ``j`` = row (y).  Every
elementwise expression is evaluated with one rounding per operation (numpy
float32, no FMA), in exactly the operand order written, so that a HIP kernel
compiled with ``-ffp-contract=off`` can be bit-identical.  Reductions over
channels are sequential in channel order (c = 0, 1, 2, ...), in float32.
"""
from __future__ import annotations

import contextlib

import numpy as np

F32 = np.float32          # the working precision; ``float64_reference()`` rebinds it to float64
F16 = np.float16


@contextlib.contextmanager
def float64_reference():
    """Evaluate every function of this module in float64 instead of float32 -- same operations, same order, same
    discrete steps (floor / clamp of the bilinear cells, the float16 casts, product and mean rounding of the DSCV,
    :276-277, which are part of the algorithm, not of the working precision).  This is the higher-precision "truth" the
    tolerance tests measure BOTH the float32 oracle and the GPU against: |gpu - f64| vs |f32 oracle - f64| says whether
    a disagreement between the two float32 evaluations is rounding noise or an error.  Not re-entrant, not thread-safe
    (test infrastructure)."""
    global F32
    old = F32
    F32 = np.float64
    try:
        yield
    finally:
        F32 = old

__all__ = [
    "get_rot_mat", "get_coords_2d", "motion_factors", "parallax2depth", "depth2parallax",
    "prev_d2para", "tile_in_batch", "interpolate_bilinear", "dense_image_warp", "back_project",
    "back_project_grad", "get_parallax_sweeping_cv", "cost_volume", "reproject",
    "recompute_depth", "normalize_cuts", "resize_bilinear_v1", "resize_nearest",
    "conv2d_same", "leaky_relu", "domain_normalization", "feature_pyramid", "disp_refiner",
    "DepthEstimatorLevel", "M4Depth", "metrics_batch", "MetricAccumulator",
    "f_input_channels", "ENCODER_CHANNELS", "REFINER_CHANNELS",
]

ENCODER_CHANNELS = [16, 32, 64, 96, 128, 192]          # m4depth_network.py:59
REFINER_CHANNELS = [128, 128, 96, 64, 32, 16, 5]       # m4depth_network.py:103,109


def _f32(x):
    return np.ascontiguousarray(x, dtype=F32)


# --------------------------------------------------------------------------
# geometry (utils/depth_operations.py)
# --------------------------------------------------------------------------
def get_rot_mat(rot):
    """utils/depth_operations.py:18-53.  rot [b,3] (small-angle xyz) or [b,4]
    (quaternion w,x,y,z, NOT renormalised) -> [b,3,3]."""
    rot = _f32(rot)
    b, c = rot.shape
    if c == 3:
        one = np.ones([b], F32)
        m = np.stack((one, -rot[:, 2], rot[:, 1],
                      rot[:, 2], one, -rot[:, 0],
                      -rot[:, 1], rot[:, 0], one), axis=-1)
        return m.reshape(b, 3, 3)
    if c == 4:
        w, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
        two = F32(2.0)
        tx = two * x
        ty = two * y
        tz = two * z
        twx = tx * w
        twy = ty * w
        twz = tz * w
        txx = tx * x
        txy = ty * x
        txz = tz * x
        tyy = ty * y
        tyz = tz * y
        tzz = tz * z
        one = F32(1.0)
        m = np.stack((one - (tyy + tzz), txy - twz, txz + twy,
                      txy + twz, one - (txx + tzz), tyz - twx,
                      txz - twy, tyz + twx, one - (txx + tyy)), axis=-1)
        return m.astype(F32).reshape(b, 3, 3)
    raise ValueError('Rotation must be expressed as a small angle (x,y,z) or a quaternion (w,x,y,z)')


def get_coords_2d(b, h, w, camera):
    """utils/depth_operations.py:57-68.  Returns coords2d [b,h,w,3] =
    ((i+.5-cx)/fx, (j+.5-cy)/fy, 1) and mesh [b,h,w,2] = (i+.5-cx, j+.5-cy)."""
    f = _f32(camera["f"]).reshape(b, 1, 1, 2)
    c = _f32(camera["c"]).reshape(b, 1, 1, 2)
    hr = np.arange(h, dtype=F32) + F32(0.5)
    wr = np.arange(w, dtype=F32) + F32(0.5)
    gx, gy = np.meshgrid(wr, hr)
    mesh = np.stack([gx, gy], axis=2).reshape(1, h, w, 2) - c
    coords = np.concatenate([mesh / f, np.ones([b, h, w, 1], F32)], axis=-1)
    return coords.astype(F32), mesh.astype(F32)


def motion_factors(b, h, w, rot, trans, camera):
    """The block shared verbatim by parallax2depth (:146-162), depth2parallax
    (:174-190) and get_parallax_sweeping_cv (:239-261) of
    utils/depth_operations.py.

    [UNPINNED] ``rot_mat @ coords2d`` is a TF batched 3x3 @ 3x1 matmul whose
    internal accumulation order is not visible in the reference; restated as
    ((R_k0*x + R_k1*y) + R_k2*1) with one rounding per operation.

    Returns dict of [b,h,w] float32 arrays: alpha, proj_x, proj_y, delta_x,
    delta_y, sqrt (= s), start_x, start_y and the per-batch scaled_t [b,3].
    """
    coords, _ = get_coords_2d(b, h, w, camera)
    R = get_rot_mat(rot)
    f = _f32(camera["f"])
    t = _f32(trans)
    x, y, one = coords[..., 0], coords[..., 1], coords[..., 2]

    def row(k):
        r0 = R[:, k, 0].reshape(b, 1, 1)
        r1 = R[:, k, 1].reshape(b, 1, 1)
        r2 = R[:, k, 2].reshape(b, 1, 1)
        return (r0 * x + r1 * y) + r2 * one

    rcx, rcy, rcz = row(0), row(1), row(2)
    fx = f[:, 0].reshape(b, 1, 1)
    fy = f[:, 1].reshape(b, 1, 1)
    alpha = rcz
    proj_x = (rcx * fx) / alpha
    proj_y = (rcy * fy) / alpha
    stx = (t[:, 0] * f[:, 0]).reshape(b, 1, 1)
    sty = (t[:, 1] * f[:, 1]).reshape(b, 1, 1)
    stz = (t[:, 2] * F32(1.0)).reshape(b, 1, 1)
    delta_x = stx - stz * proj_x
    delta_y = sty - stz * proj_y
    s = np.sqrt(delta_x * delta_x + delta_y * delta_y)
    return dict(alpha=alpha, proj_x=proj_x, proj_y=proj_y, delta_x=delta_x, delta_y=delta_y,
                sqrt=s, start_x=x * fx, start_y=y * fy, stz=stz)


def parallax2depth(disp, rot, trans, camera):
    """utils/depth_operations.py:141-166: depth = (s/disp - tz)/alpha."""
    disp = _f32(disp)
    b, h, w = disp.shape[:3]
    m = motion_factors(b, h, w, rot, trans, camera)
    depth = (m["sqrt"] / disp[..., 0] - m["stz"]) / m["alpha"]
    return depth.astype(F32)[..., None]


def depth2parallax(depth, rot, trans, camera):
    """utils/depth_operations.py:169-194: disp = s/(depth*alpha + tz)."""
    depth = _f32(depth)
    b, h, w = depth.shape[:3]
    m = motion_factors(b, h, w, rot, trans, camera)
    disp = m["sqrt"] / (depth[..., 0] * m["alpha"] + m["stz"])
    return disp.astype(F32)[..., None]


def prev_d2para(prev_d, rot, trans, camera):
    """utils/depth_operations.py:197-215 (rot is unused by the reference).
    delta = (t*f - tz*((mesh/f)*f)) / (prev_d - tz); para = sqrt(dx^2+dy^2)."""
    prev_d = _f32(prev_d)
    b, h, w = prev_d.shape[:3]
    coords, _ = get_coords_2d(b, h, w, camera)
    f = _f32(camera["f"])
    t = _f32(trans)
    fx = f[:, 0].reshape(b, 1, 1)
    fy = f[:, 1].reshape(b, 1, 1)
    cx = coords[..., 0] * fx
    cy = coords[..., 1] * fy
    stx = (t[:, 0] * f[:, 0]).reshape(b, 1, 1)
    sty = (t[:, 1] * f[:, 1]).reshape(b, 1, 1)
    tz = t[:, 2].reshape(b, 1, 1)
    den = prev_d[..., 0] - tz
    dx = (stx - tz * cx) / den
    dy = (sty - tz * cy) / den
    para = np.sqrt(dx * dx + dy * dy)      # tf.norm(axis=2) over the 2 components
    return para.astype(F32)[..., None]


def tile_in_batch(m, nbre_copies):
    """utils/depth_operations.py:217-221: out batch index = copy*b + bi."""
    m = np.asarray(m)
    return np.tile(m[None], [nbre_copies] + [1] * m.ndim).reshape((-1,) + m.shape[1:])


# --------------------------------------------------------------------------
# bilinear warp (utils/dense_image_warp.py) and the BackProject op
# --------------------------------------------------------------------------
def interpolate_bilinear(grid, query, return_index=False):
    """utils/dense_image_warp.py:61-192 with indexing='ij' (the TF-CPU path).
    grid [B,H,W,C], query [B,N,2] as (row, col).  Returns [B,N,C]; with
    return_index also the int32 floors (y0, x0) [B,N] (bit-exact contract)."""
    grid = _f32(grid)
    query = _f32(query)
    B, H, W, C = grid.shape
    alphas, floors = [], []
    for dim, size in ((0, H), (1, W)):
        q = query[..., dim]
        max_floor = F32(size - 2)
        fl = np.minimum(np.maximum(F32(0.0), np.floor(q)), max_floor)     # :137-138
        floors.append(fl.astype(np.int32))                                  # :139
        a = q - fl                                                          # :146
        a = np.minimum(np.maximum(F32(0.0), a), F32(1.0))                   # :149
        alphas.append(a[..., None])
    y0, x0 = floors
    y1, x1 = y0 + 1, x0 + 1
    flat = grid.reshape(B * H * W, C)
    boff = (np.arange(B, dtype=np.int64) * H * W).reshape(B, 1)

    def gather(yc, xc):
        return flat[boff + yc.astype(np.int64) * W + xc]

    tl, tr, bl, br = gather(y0, x0), gather(y0, x1), gather(y1, x0), gather(y1, x1)
    top = alphas[1] * (tr - tl) + tl                                        # :188
    bot = alphas[1] * (br - bl) + bl                                        # :189
    out = (alphas[0] * (bot - top) + top).astype(F32)                       # :190
    if return_index:
        return out, y0, x0
    return out


def back_project(inputs, coords):
    """cuda_backproject/backproject_op_gpu.cu.cc:19-79 (BackProjectForward).
    inputs [B,H,W,F,C], coords [B,H,W,S,F,2] as (x, y) -> [B,H,W,S,F,C].
    Out-of-image coordinates give 0 (the op memsets its output, :91)."""
    inputs = _f32(inputs)
    coords = _f32(coords)
    B, H, W, Fd, C = inputs.shape
    S = coords.shape[3]
    x = coords[..., 0]
    y = coords[..., 1]
    inside = (x >= 0) & (y >= 0) & (x <= F32(W - 1)) & (y <= F32(H - 1))
    xs = np.where(inside, x, F32(0))
    ys = np.where(inside, y, F32(0))
    x0 = np.floor(xs).astype(np.int32)
    x1 = np.ceil(xs).astype(np.int32)
    y0 = np.floor(ys).astype(np.int32)
    y1 = np.ceil(ys).astype(np.int32)
    dx = xs - x0.astype(F32)
    dy = ys - y0.astype(F32)
    one = F32(1)
    w00 = (one - dy) * (one - dx)
    w01 = (one - dy) * dx
    w10 = dy * (one - dx)
    w11 = dy * dx
    n = np.arange(B).reshape(B, 1, 1, 1, 1)
    fi = np.arange(Fd).reshape(1, 1, 1, 1, Fd)
    n, fi = np.broadcast_arrays(n, fi, x0)[:2]
    im00 = inputs[n, y0, x0, fi]
    im01 = inputs[n, y0, x1, fi]
    im10 = inputs[n, y1, x0, fi]
    im11 = inputs[n, y1, x1, fi]
    # ((im00*w00 + im01*w01) + im10*w10) + im11*w11, one rounding per op (:73)
    out = ((im00 * w00[..., None] + im01 * w01[..., None]) + im10 * w10[..., None]) + im11 * w11[..., None]
    out = np.where(inside[..., None], out, F32(0)).astype(F32)
    return out.reshape(B, H, W, S, Fd, C)


def back_project_grad(inputs, coords, grad):
    """cuda_backproject/backproject_op_gpu.cu.cc:108-197 (BackProjectBackward).
    Returns (inputs_grad [B,H,W,F,C], coords_grad [B,H,W,S,F,2]).  The scatter
    into inputs_grad is an fp32 atomicAdd in the reference, i.e. its summation
    order is undefined; accumulated here in float64 and rounded once, so tests
    compare with a tolerance, not bit-exactly."""
    inputs = _f32(inputs)
    coords = _f32(coords)
    grad = _f32(grad)
    B, H, W, Fd, C = inputs.shape
    S = coords.shape[3]
    x = coords[..., 0]
    y = coords[..., 1]
    inside = (x >= 0) & (y >= 0) & (x <= F32(W - 1)) & (y <= F32(H - 1))
    xs = np.where(inside, x, F32(0))
    ys = np.where(inside, y, F32(0))
    x0 = np.floor(xs).astype(np.int32)
    x1 = np.ceil(xs).astype(np.int32)
    y0 = np.floor(ys).astype(np.int32)
    y1 = np.ceil(ys).astype(np.int32)
    dx = xs - x0.astype(F32)
    dy = ys - y0.astype(F32)
    one = F32(1)
    wx0, wx1, wy0, wy1 = one - dx, dx, one - dy, dy
    w = [(one - dy) * (one - dx), (one - dy) * dx, dy * (one - dx), dy * dx]
    n = np.arange(B).reshape(B, 1, 1, 1, 1)
    fi = np.arange(Fd).reshape(1, 1, 1, 1, Fd)
    n, fi = np.broadcast_arrays(n, fi, x0)[:2]
    corners = [(y0, x0), (y0, x1), (y1, x0), (y1, x1)]
    g = np.where(inside[..., None], grad, F32(0))
    acc = np.zeros(inputs.shape, np.float64)
    for (yy, xx), ww in zip(corners, w):
        np.add.at(acc, (n, yy, xx, fi), (g * ww[..., None]).astype(np.float64))
    im00, im01, im10, im11 = [inputs[n, yy, xx, fi] for (yy, xx) in corners]
    gx = np.zeros(x.shape, F32)
    gy = np.zeros(x.shape, F32)
    for c in range(C):      # sequential over channels like the kernel's loop (:171-186)
        gc = g[..., c]
        gx = gx + gc * (wy0 * (im01[..., c] - im00[..., c]) + wy1 * (im11[..., c] - im10[..., c]))
        gy = gy + gc * (wx0 * (im10[..., c] - im00[..., c]) + wx1 * (im11[..., c] - im01[..., c]))
    cg = np.stack([gx, gy], axis=-1).astype(F32)
    return acc.astype(F32), cg


def dense_image_warp(image, flow, use_backproject=False, return_index=False):
    """utils/dense_image_warp.py:195-268.  image [B,H,W,C], flow [B,H,W,2]
    in (row, col) order; query = grid + flow (:244 -- PLUS, despite the
    docstring).  Default is the TF-CPU branch (:254-259); use_backproject
    follows the CUDA-op branch (:246-253)."""
    image = _f32(image)
    flow = _f32(flow)
    B, H, W, C = image.shape
    gx, gy = np.meshgrid(np.arange(W), np.arange(H))
    grid = np.stack([gy, gx], axis=2).astype(F32)[None]
    query = grid + flow
    if use_backproject:
        lo = np.array([0.0, 0.0], F32)
        hi = np.array([H - 1, W - 1], F32)
        q = np.minimum(np.maximum(query, lo), hi)
        coords = q[..., ::-1].reshape(B, H, W, 1, 1, 2)
        out = back_project(image.reshape(B, H, W, 1, C), coords)
        return out.reshape(B, H, W, C)
    res = interpolate_bilinear(image, query.reshape(B, H * W, 2), return_index=return_index)
    if return_index:
        out, y0, x0 = res
        return out.reshape(B, H, W, C), y0.reshape(B, H, W), x0.reshape(B, H, W)
    return res.reshape(B, H, W, C)


def reproject(map_, depth, rot, trans, camera):
    """utils/depth_operations.py:72-105.  Full 6-DoF projective warp.
    [UNPINNED] the two TF matmuls (3x3 @ 3x4, then 3x4 @ 4x1) are restated as
    left-to-right sequential dot products."""
    map_ = _f32(map_)
    depth = _f32(depth)
    b, h, w, _ = map_.shape
    if depth.shape[1] != h or depth.shape[2] != w:
        raise ValueError('Height and width of map and depth should be the same')
    f = _f32(camera["f"])
    R = get_rot_mat(rot)
    t = _f32(trans)
    T = np.concatenate([R, t[:, :, None]], axis=-1)                  # [b,3,4]
    P = np.zeros([b, 3, 3], F32)
    P[:, 0, 0] = f[:, 0]
    P[:, 1, 1] = f[:, 1]
    P[:, 2, 2] = 1.0
    M = np.zeros([b, 3, 4], F32)
    for r in range(3):
        for c in range(4):
            acc = P[:, r, 0] * T[:, 0, c]
            acc = acc + P[:, r, 1] * T[:, 1, c]
            acc = acc + P[:, r, 2] * T[:, 2, c]
            M[:, r, c] = acc
    coords, mesh = get_coords_2d(b, h, w, camera)
    pos = np.concatenate([coords * depth, np.ones([b, h, w, 1], F32)], axis=-1)   # [b,h,w,4]

    def mv(ncols):
        out = []
        for r in range(3):
            acc = M[:, r, 0].reshape(b, 1, 1) * pos[..., 0]
            for c in range(1, ncols):
                acc = acc + M[:, r, c].reshape(b, 1, 1) * pos[..., c]
            out.append(acc)
        return out

    px, py, pz = mv(4)
    rx, ry, rz = mv(3)
    proj = np.stack([px / pz, py / pz], axis=-1)
    rotc = np.stack([rx / rz, ry / rz], axis=-1)
    flow = (proj - mesh)[..., ::-1]
    return dense_image_warp(map_, flow), [(proj - rotc).astype(F32), rotc.astype(F32)]


def recompute_depth(depth, rot, trans, camera, mesh=None):
    """utils/depth_operations.py:109-137 (exported, never called by the
    reference): new_depth = clip((R_z . coords2d)*depth + (R_z . -t), .1, 2000)."""
    depth = _f32(depth)
    b, h, w, _ = depth.shape
    R = get_rot_mat(rot)[:, 2, :]
    t = -_f32(trans)
    if mesh is None:
        _, mesh = get_coords_2d(b, h, w, camera)
    f = _f32(camera["f"]).reshape(b, 1, 1, 2)
    c2 = mesh / f
    r0, r1, r2 = [R[:, k].reshape(b, 1, 1) for k in range(3)]
    tv = ((R[:, 0] * t[:, 0] + R[:, 1] * t[:, 1]) + R[:, 2] * t[:, 2]).reshape(b, 1, 1)
    pr = (r0 * c2[..., 0] + r1 * c2[..., 1]) + r2 * F32(1)
    nd = pr * depth[..., 0] + tv
    return np.clip(nd, F32(0.1), F32(2000.)).astype(F32)[..., None]


# --------------------------------------------------------------------------
# cost volumes
# --------------------------------------------------------------------------
def get_parallax_sweeping_cv(c1, c2, disp_prev_t, disp, rot, trans, camera, search_range,
                             nbre_cuts=1, cv_accum="fp32_round", return_index=False):
    """DSCV, utils/depth_operations.py:224-281.

    c1, c2 [b,h,w,C]; disp_prev_t, disp [b,h,w,1].  Returns cv [b,h,w,k*(2r+1)]
    (cut-major: channel = kk*(2r+1) + (n+r), from the reshape/transpose at :278)
    and prev_disp [b,h,w,2r+1].

    [UNPINNED] ``tf.reduce_mean`` over float16 (:277): Eigen may accumulate in
    half sequentially, XLA-CPU (enabled by main.py:23-24 for eval) may
    accumulate in float.  cv_accum = "fp32_round" (default; fp16 products summed
    sequentially in float32, divided by n in float32, rounded once to half) or
    "fp16_seq" (sequential half adds, half divide)."""
    c1 = _f32(c1)
    c2 = _f32(c2)
    disp = _f32(disp)
    disp_prev_t = _f32(disp_prev_t)
    b, h, w, C = c1.shape
    r = int(search_range)
    ncp = 2 * r + 1
    k = int(nbre_cuts)
    n_c = C // k
    m = motion_factors(b, h, w, rot, trans, camera)
    src = np.concatenate([c2, disp_prev_t], axis=-1)                       # :268
    c1h = c1.astype(F16)
    cv = np.zeros([b, h, w, k * ncp], F32)
    prev_disp = np.zeros([b, h, w, ncp], F32)
    jj, ii = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    jj = jj.astype(F32)[None]
    ii = ii.astype(F32)[None]
    idx = []
    for t in range(ncp):
        n = F32(t - r)
        p = np.minimum(np.maximum(disp[..., 0] + n, F32(1e-6)), F32(1e6))  # :235-236
        divider = m["sqrt"] / p                                            # :262
        dxx = m["delta_x"] / divider                                       # :263
        dyy = m["delta_y"] / divider
        flow_x = (m["proj_x"] + dxx) - m["start_x"]                        # :264
        flow_y = (m["proj_y"] + dyy) - m["start_y"]
        query = np.stack([jj + flow_y, ii + flow_x], axis=-1).reshape(b, h * w, 2)
        wv, y0, x0 = interpolate_bilinear(src, query, return_index=True)
        wv = wv.reshape(b, h, w, C + 1)
        idx.append((y0.reshape(b, h, w), x0.reshape(b, h, w)))
        prev_disp[..., t] = wv[..., C]
        prod = c1h * wv[..., :C].astype(F16)                               # :276 (half * half -> half)
        for kk in range(k):
            pc = prod[..., kk * n_c:(kk + 1) * n_c]
            if cv_accum == "fp32_round":
                acc = pc[..., 0].astype(F32)
                for c in range(1, n_c):
                    acc = acc + pc[..., c].astype(F32)
                val = (acc / F32(n_c)).astype(F16)
            elif cv_accum == "fp16_seq":
                acc = pc[..., 0]
                for c in range(1, n_c):
                    acc = (acc + pc[..., c]).astype(F16)
                val = (acc / F16(n_c)).astype(F16)
            else:
                raise ValueError(cv_accum)
            cv[..., kk * ncp + t] = val.astype(F32)
    if return_index:
        y0 = np.stack([a for a, _ in idx], axis=-1)
        x0 = np.stack([a for _, a in idx], axis=-1)
        return cv, prev_disp, y0, x0
    return cv, prev_disp


def leaky_relu(x, alpha=0.1):
    """tf.nn.leaky_relu: x if x > 0 else alpha*x."""
    x = _f32(x)
    return np.where(x > 0, x, x * F32(alpha)).astype(F32)


def cost_volume(c1, c2, search_range, name="cost_volume", dilation_rate=1, nbre_cuts=1):
    """SNCV, utils/depth_operations.py:284-313.  Output channel
    ((y*(2r+1)+x)*k + kk) = leaky_relu(mean_c c1[j,i,c]*c2pad[j+y*d, i+x*d, c]).
    [UNPINNED] reduce_mean order: sequential float32 sum over the cut's
    channels, then a float32 divide by the channel count."""
    c1 = _f32(c1)
    c2 = _f32(c2)
    b, h, w, C = c1.shape
    r = int(search_range)
    d = int(dilation_rate)
    k = int(nbre_cuts)
    n_c = C // k
    sr = r * d
    pad = np.zeros([b, h + 2 * sr, w + 2 * sr, C], F32)
    pad[:, sr:sr + h, sr:sr + w, :] = c2
    mo = 2 * r + 1
    out = np.zeros([b, h, w, mo * mo * k], F32)
    for y in range(mo):
        for x in range(mo):
            sl = pad[:, y * d:y * d + h, x * d:x * d + w, :]
            prod = c1 * sl
            for kk in range(k):
                pc = prod[..., kk * n_c:(kk + 1) * n_c]
                acc = pc[..., 0].copy()
                for c in range(1, n_c):
                    acc = acc + pc[..., c]
                out[..., (y * mo + x) * k + kk] = acc / F32(n_c)
    return leaky_relu(out, 0.1)


# --------------------------------------------------------------------------
# level glue + host-side network pieces (m4depth_network.py)
# --------------------------------------------------------------------------
def normalize_cuts(x, nbre_cuts):
    """m4depth_network.py:179-189: tf.linalg.normalize over each cut's
    channels, x / sqrt(sum x^2) (no epsilon).  [UNPINNED] sum order: sequential."""
    x = _f32(x)
    b, h, w, C = x.shape
    n_c = C // nbre_cuts
    out = np.empty_like(x)
    for kk in range(nbre_cuts):
        xc = x[..., kk * n_c:(kk + 1) * n_c]
        acc = xc[..., 0] * xc[..., 0]
        for c in range(1, n_c):
            acc = acc + xc[..., c] * xc[..., c]
        nrm = np.sqrt(acc)
        with np.errstate(invalid="ignore", divide="ignore"):
            out[..., kk * n_c:(kk + 1) * n_c] = xc / nrm[..., None]
    return out


def resize_bilinear_v1(x, out_h, out_w):
    """tf.compat.v1.image.resize_bilinear defaults (align_corners=False,
    half_pixel_centers=False), as called at m4depth_network.py:202-204.
    src = dst*(in/out); lower=floor, upper=min(ceil, in-1), lerp=src-floor;
    top = tl + (tr-tl)*xl; bot = bl + (br-bl)*xl; out = top + (bot-top)*yl."""
    x = _f32(x)
    b, ih, iw, c = x.shape

    def weights(out_n, in_n):
        scale = F32(in_n) / F32(out_n)
        src = np.arange(out_n, dtype=F32) * scale
        fl = np.floor(src)
        lo = np.maximum(fl.astype(np.int64), 0)
        hi = np.minimum(np.ceil(src).astype(np.int64), in_n - 1)
        return lo, hi, (src - fl).astype(F32)

    ylo, yhi, yl = weights(out_h, ih)
    xlo, xhi, xl = weights(out_w, iw)
    xl = xl.reshape(1, 1, out_w, 1)
    yl = yl.reshape(1, out_h, 1, 1)
    tl = x[:, ylo][:, :, xlo]
    tr = x[:, ylo][:, :, xhi]
    bl = x[:, yhi][:, :, xlo]
    br = x[:, yhi][:, :, xhi]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return (top + (bot - top) * yl).astype(F32)


def resize_nearest(x, out_h, out_w):
    """tf.image.resize(..., NEAREST_NEIGHBOR) (v2: half-pixel centres), as at
    m4depth_network.py:368: src = min(floor((dst+0.5)*in/out), in-1)."""
    x = _f32(x)
    ih, iw = x.shape[1:3]
    ys = np.minimum(np.floor((np.arange(out_h, dtype=F32) + F32(0.5)) * (F32(ih) / F32(out_h))).astype(np.int64), ih - 1)
    xs = np.minimum(np.floor((np.arange(out_w, dtype=F32) + F32(0.5)) * (F32(iw) / F32(out_w))).astype(np.int64), iw - 1)
    return x[:, ys][:, :, xs]


def conv2d_same(x, kernel, bias, stride):
    """Keras Conv2D(3, padding='same') (m4depth_network.py:63-72,104-114).
    kernel HWIO [3,3,Cin,Cout].  TF SAME: out = ceil(in/stride), total pad =
    max((out-1)*stride + 3 - in, 0), pad_before = total//2 (so stride 2 on an
    even size pads 0 before / 1 after).  Accumulates the 9 taps in (ky,kx)
    order, each tap a float32 BLAS matmul -- host-side conv, not hot path."""
    x = _f32(x)
    b, h, w, cin = x.shape
    kh, kw, _, cout = kernel.shape
    oh = -(-h // stride)
    ow = -(-w // stride)
    ph = max((oh - 1) * stride + kh - h, 0)
    pw = max((ow - 1) * stride + kw - w, 0)
    pt, pl = ph // 2, pw // 2
    xp = np.zeros([b, h + ph, w + pw, cin], F32)
    xp[:, pt:pt + h, pl:pl + w] = x
    out = np.zeros([b, oh, ow, cout], F32)
    for ky in range(kh):
        for kx in range(kw):
            sl = xp[:, ky:ky + (oh - 1) * stride + 1:stride, kx:kx + (ow - 1) * stride + 1:stride, :]
            out += (sl.reshape(-1, cin) @ _f32(kernel[ky, kx])).reshape(b, oh, ow, cout)
    if bias is not None:
        out = out + _f32(bias)
    return out.astype(F32)


def domain_normalization(x, scale, bias):
    """m4depth_network.py:44-48 -- note (x-mean)/(var+1e-12): var, not std."""
    x = _f32(x)
    mean = x.mean(axis=(1, 2), keepdims=True, dtype=F32)
    var = ((x - mean) * (x - mean)).mean(axis=(1, 2), keepdims=True, dtype=F32)
    n = (x - mean) / (var + F32(1e-12))
    ss = (n * n).sum(axis=-1, keepdims=True, dtype=F32)
    n = n * (F32(1.0) / np.sqrt(np.maximum(ss, F32(1e-12))))           # tf.math.l2_normalize
    return (_f32(scale).reshape(1, 1, 1, -1) * n + _f32(bias).reshape(1, 1, 1, -1)).astype(F32)


def feature_pyramid(images, weights, nbre_lvls, use_dinl=True):
    """FeaturePyramid.call, m4depth_network.py:76-90.  Returns fine->coarse."""
    fm = _f32(images)
    outs = []
    for i in range(nbre_lvls):
        t = conv2d_same(fm, weights[f"enc.s1.{i}.kernel"], weights[f"enc.s1.{i}.bias"], 1)
        if use_dinl and i == 0:
            t = domain_normalization(t, weights["enc.dn.0.scale"], weights["enc.dn.0.bias"])
        t = leaky_relu(t, 0.1)
        t = conv2d_same(t, weights[f"enc.s2.{i}.kernel"], weights[f"enc.s2.{i}.bias"], 2)
        fm = leaky_relu(t, 0.1)
        outs.append(fm)
    return outs


def disp_refiner(f_input, weights, lvl):
    """DispRefiner.call, m4depth_network.py:116-135 (first returned tensor)."""
    x = _f32(f_input)
    n = len(REFINER_CHANNELS)
    for i in range(n):
        x = conv2d_same(x, weights[f"lvl.{lvl}.conv.{i}.kernel"], weights[f"lvl.{lvl}.conv.{i}.bias"], 1)
        if i < n - 1:
            x = leaky_relu(x, 0.1)
    return x


def f_input_channels(nbre_cuts, dscv_range=4, sncv_range=3, ablation=None):
    a = ablation or {}
    n = (2 * dscv_range + 1) * nbre_cuts + 1
    if a.get("level_memory", True):
        n += 4
    if a.get("SNCV", True):
        n += (2 * sncv_range + 1) ** 2 * nbre_cuts
    if a.get("time_recurr", True):
        n += 1
    return n


DEFAULT_ABLATION = dict(DINL=True, SNCV=True, time_recurr=True, normalize_features=True,
                        subdivide_features=True, level_memory=True)


class DepthEstimatorLevel:
    """m4depth_network.py:138-262, inference mode (is_training=False): owns
    the recurrent state prev_f_maps / depth_prev_t.  ``dscv_range`` and
    ``sncv_range`` are hard-coded to 4 and 3 in the reference (:221, :232)."""

    def __init__(self, weights, depth, ablation=None, dscv_range=4, sncv_range=3, cv_accum="fp32_round"):
        self.weights = weights
        self.lvl_depth = depth
        self.lvl_mul = depth - 3
        self.ablation = dict(DEFAULT_ABLATION, **(ablation or {}))
        self.dscv_range = dscv_range
        self.sncv_range = sncv_range
        self.cv_accum = cv_accum
        self.prev_f_maps = None
        self.depth_prev_t = None
        self.last_f_input = None

    def nbre_cuts(self):
        return 2 ** (self.lvl_depth // 2) if self.ablation["subdivide_features"] else 1

    def __call__(self, curr_f_maps, prev_l_est, rot, trans, camera, new_traj):
        curr_f_maps = _f32(curr_f_maps)
        b, h, w, c = curr_f_maps.shape
        k = self.nbre_cuts()
        if self.ablation["normalize_features"]:
            curr_f = normalize_cuts(curr_f_maps, k)                            # :179-186
        else:
            curr_f = curr_f_maps
        if self.prev_f_maps is None:                                           # build(): :157-163
            self.prev_f_maps = np.zeros([b, h, w, c], F32)
            self.depth_prev_t = np.ones([b, h, w, 1], F32)
        if prev_l_est is None:                                                 # :196-200
            para_prev_l = np.ones([b, h, w, 1], F32)
            depth_prev_l = np.full([b, h, w, 1], 1000., F32)
            other_prev_l = np.zeros([b, h, w, 4], F32)
        else:                                                                  # :202-204
            other_prev_l = resize_bilinear_v1(prev_l_est["other"], h, w)
            para_prev_l = resize_bilinear_v1(prev_l_est["parallax"], h, w) * F32(2.)
            depth_prev_l = resize_bilinear_v1(prev_l_est["depth"], h, w)
        if bool(np.asarray(new_traj).reshape(-1)[0]):                          # :208-214
            self.prev_f_maps = curr_f
            self.depth_prev_t = np.full([b, h, w, 1], 1000., F32)
            return {"depth": depth_prev_l, "parallax": para_prev_l, "other": other_prev_l}
        para_prev_t = prev_d2para(self.depth_prev_t, rot, trans, camera)       # :218
        cv, para_reproj = get_parallax_sweeping_cv(curr_f, self.prev_f_maps, para_prev_t, para_prev_l,
                                                   rot, trans, camera, self.dscv_range, nbre_cuts=k,
                                                   cv_accum=self.cv_accum)     # :220-221
        scale = F32(2.0 ** self.lvl_mul)
        feats = [cv, np.log(para_prev_l * scale)]                              # :224
        if self.ablation["level_memory"]:
            feats.append(other_prev_l)
        if self.ablation["SNCV"]:
            feats.append(cost_volume(curr_f, curr_f, self.sncv_range, nbre_cuts=k))   # :232
        if self.ablation["time_recurr"]:
            r = self.dscv_range
            feats.append(np.log(para_reproj[..., r:r + 1] * scale))            # :238 ([...,4:5] at r=4)
        f_input = np.concatenate(feats, axis=3).astype(F32)
        self.last_f_input = f_input
        out = disp_refiner(f_input, self.weights, self.lvl_depth)              # :245
        para = out[..., :1]
        other = out[..., 1:]
        para_curr = (np.exp(np.minimum(np.maximum(para, F32(-7.)), F32(7.))) / scale).astype(F32)   # :250
        depth = parallax2depth(para_curr, rot, trans, camera)                  # :251
        self.prev_f_maps = curr_f                                              # :258-260
        self.depth_prev_t = depth
        return {"other": other.astype(F32), "depth": depth, "parallax": para_curr}


class M4Depth:
    """M4Depth.call + DepthEstimatorPyramid.call in inference mode
    (m4depth_network.py:278-323, 351-369)."""

    def __init__(self, weights, nbre_levels=6, ablation=None, dscv_range=4, sncv_range=3, cv_accum="fp32_round"):
        self.weights = weights
        self.nbre_levels = nbre_levels
        self.ablation = dict(DEFAULT_ABLATION, **(ablation or {}))
        self.levels = [DepthEstimatorLevel(weights, i + 1, self.ablation, dscv_range, sncv_range, cv_accum)
                       for i in range(nbre_levels)]

    def __call__(self, traj_samples, camera):
        L = self.nbre_levels
        pyrs = [feature_pyramid(s["RGB_im"], self.weights, L, self.ablation["DINL"]) for s in traj_samples]
        d_est_seq = []
        for f_pyr, sample in zip(pyrs, traj_samples):
            cnter = float(L)
            d_est_curr = None
            for l in range(L):                                                 # coarse -> fine (:293)
                lvl = L - 1 - l
                cam = {"f": _f32(camera["f"]) / F32(2. ** cnter), "c": _f32(camera["c"]) / F32(2. ** cnter)}
                prev = None if d_est_curr is None else dict(d_est_curr[-1])
                est = self.levels[lvl](f_pyr[lvl], prev, sample["rot"], sample["trans"], cam, sample["new_traj"])
                d_est_curr = [est] if d_est_curr is None else d_est_curr + [est]
                cnter -= 1.
            d_est_seq.append(d_est_curr[::-1])
        h, w = traj_samples[-1]["RGB_im"].shape[1:3]
        return {"depth": resize_nearest(d_est_seq[-1][0]["depth"], h, w)}, d_est_seq


# --------------------------------------------------------------------------
# metrics (metrics.py) via test_step (m4depth_network.py:462-470)
# --------------------------------------------------------------------------
METRIC_NAMES = ["AbsRel", "SqRel", "RMSE", "RMSE_log", "Delta1", "Delta2", "Delta3"]


def _mrm(err, ref):
    """metrics.py:3-5 masked_reduce_mean (mask on ref > 1e-6, no-nan multiply)."""
    mask = (ref > F32(1e-6)).astype(F32)
    e = np.where(mask != 0, err, F32(0)).astype(np.float64)       # multiply_no_nan
    return F32(e.sum() / max(float(mask.sum(dtype=np.float64)), 1.0))


def metrics_batch(gt, est, max_d=80.):
    """One test_step update: clip gt to [0,max_d], est to [0.001,max_d]
    (m4depth_network.py:465-467), then the 7 per-batch scalars in main.py:127-130
    order.  Sums are float64 here (TF sums float32 with an unspecified tree
    order) -- compare with a tolerance."""
    gt = np.clip(_f32(gt), F32(0.0), F32(max_d))
    est = np.clip(_f32(est), F32(0.001), F32(max_d))
    with np.errstate(divide="ignore", invalid="ignore"):
        absrel = _mrm(np.abs(gt - est) / (gt + F32(1e-6)), gt)
        sqrel = _mrm((gt - est) * (gt - est) / (gt + F32(1e-6)), gt)
        rmse = F32(np.sqrt(_mrm((gt - est) * (gt - est), gt)))
        lg = np.log(gt + F32(1e-6))
        le = np.log(est + F32(1e-6))
        rmsel = F32(np.sqrt(_mrm((lg - le) * (lg - le), lg)))      # mask on the LOG (metrics.py:24-28)
        th = np.maximum(gt / est, est / gt)
        deltas = [_mrm((th < F32(1.25 ** n)).astype(F32), gt) for n in (1, 2, 3)]
    return np.array([absrel, sqrel, rmse, rmsel] + deltas, F32)


class MetricAccumulator:
    """Keras ``Mean`` semantics: total += per-batch scalar, count += 1."""

    def __init__(self):
        self.total = np.zeros(7, np.float64)
        self.count = 0

    def update(self, gt, est):
        self.total += metrics_batch(gt, est).astype(np.float64)
        self.count += 1

    def result(self):
        return (self.total / max(self.count, 1)).astype(F32)
