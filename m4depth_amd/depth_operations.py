"""Geometry and cost-volume ops -- MI355X native.

Same names, argument order and error behaviour as the reference module
``utils/depth_operations.py``; tensors are ``torch.Tensor`` on the ROCm device
(NHWC float32), ``camera`` is the dict ``{"f": [b,2], "c": [b,2]}`` of
level-local intrinsics.  Every map-sized computation runs in a HIP kernel of
libm4depth_hip.so (no PyTorch/CPU implementation of the hot path exists here);
only the tiny per-sample 3x3 rotation helper is plain tensor plumbing.
"""
from __future__ import annotations

import torch

from .dense_image_warp import dense_image_warp
from ._lib import lib, dptr, stream_ptr, check, as_f32


__all__ = ["wrap_feature_block", "get_rot_mat", "get_coords_2d", "reproject", "recompute_depth",
           "parallax2depth", "depth2parallax", "prev_d2para", "tile_in_batch",
           "get_parallax_sweeping_cv", "cost_volume", "dense_image_warp"]


def _motion_args(rot, trans, camera, b, need_rot=True):
    if need_rot:
        rot = as_f32(rot, "rot")
        if rot.dim() != 2 or rot.shape[1] not in (3, 4):
            raise ValueError('Rotation must be expressed as a small angle (x,y,z) or a quaternion (w,x,y,z)')
    trans = as_f32(trans, "trans").reshape(b, 3)
    f = as_f32(camera["f"], "camera['f']").reshape(b, 2)
    c = as_f32(camera["c"], "camera['c']").reshape(b, 2)
    return rot, trans, f, c


def get_rot_mat(rot):
    """utils/depth_operations.py:18-53: [b,3] small-angle xyz or [b,4]
    quaternion (w,x,y,z) -> [b,3,3].  Per-sample scalars: host-side plumbing."""
    if rot.dim() != 2:
        raise ValueError('Rotation must be expressed as a small angle (x,y,z) or a quaternion (w,x,y,z)')
    b, c = rot.shape
    if c == 3:
        ones = torch.ones([b], dtype=rot.dtype, device=rot.device)
        m = torch.stack((ones, -rot[:, 2], rot[:, 1], rot[:, 2], ones, -rot[:, 0],
                         -rot[:, 1], rot[:, 0], ones), dim=-1)
        return m.reshape(b, 3, 3)
    if c == 4:
        w, x, y, z = rot.unbind(-1)
        tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
        twx, twy, twz = tx * w, ty * w, tz * w
        txx, txy, txz = tx * x, ty * x, tz * x
        tyy, tyz, tzz = ty * y, tz * y, tz * z
        m = torch.stack((1.0 - (tyy + tzz), txy - twz, txz + twy,
                         txy + twz, 1.0 - (txx + tzz), tyz - twx,
                         txz - twy, tyz + twx, 1.0 - (txx + tyy)), dim=-1)
        return m.reshape(b, 3, 3)
    raise ValueError('Rotation must be expressed as a small angle (x,y,z) or a quaternion (w,x,y,z)')


def get_coords_2d(map, camera):
    """utils/depth_operations.py:57-68 -> (coords_2d [b,h,w,3,1], mesh [b,h,w,2]).
    The kernels recompute this grid in registers; this tensor form exists for API
    parity only."""
    b, h, w = map.shape[:3]
    dev = map.device
    h_range = torch.arange(0., h, 1.0, dtype=torch.float32, device=dev) + 0.5
    w_range = torch.arange(0., w, 1.0, dtype=torch.float32, device=dev) + 0.5
    grid_y, grid_x = torch.meshgrid(h_range, w_range, indexing="ij")
    mesh = torch.stack([grid_x, grid_y], dim=2).reshape(1, h, w, 2) - camera["c"].reshape(b, 1, 1, 2)
    coords_2d = torch.cat([mesh / camera["f"].reshape(b, 1, 1, 2),
                           torch.ones([b, h, w, 1], dtype=torch.float32, device=dev)], dim=-1)
    return coords_2d.unsqueeze(-1), mesh


def _converter(fn, name, x, rot, trans, camera, need_rot=True):
    x = as_f32(x, name)
    b, h, w = x.shape[:3]
    rot, trans, f, c = _motion_args(rot, trans, camera, b, need_rot)
    out = torch.empty((b, h, w, 1), dtype=torch.float32, device=x.device)
    rot_c = rot.shape[1] if need_rot else 0
    check(fn(dptr(x, name), dptr(rot, "rot") if need_rot else None, rot_c, dptr(trans, "trans"),
             dptr(f, "camera['f']"), dptr(c, "camera['c']"), b, h, w, dptr(out), stream_ptr()), fn.__name__)
    return out


def parallax2depth(disp, rot, trans, camera):
    """utils/depth_operations.py:141-166."""
    return _converter(lib.m4d_parallax2depth, "disp", disp, rot, trans, camera)


def depth2parallax(depth, rot, trans, camera):
    """utils/depth_operations.py:169-194."""
    return _converter(lib.m4d_depth2parallax, "depth", depth, rot, trans, camera)


def prev_d2para(prev_d, rot, trans, camera):
    """utils/depth_operations.py:197-215 (``rot`` is unused, as in the reference)."""
    return _converter(lib.m4d_prev_d2para, "prev_d", prev_d, rot, trans, camera, need_rot=False)


def recompute_depth(depth, rot, trans, camera, mesh=None):
    """utils/depth_operations.py:109-137.  ``mesh`` may only be the default grid."""
    if mesh is not None:
        raise NotImplementedError("recompute_depth: a custom mesh is not supported (the reference never passes one)")
    return _converter(lib.m4d_recompute_depth, "depth", depth, rot, trans, camera)


def reproject(map, depth, rot, trans, camera):
    """utils/depth_operations.py:72-105 -> (warped map, [proj - rot_coord, rot_coord])."""
    map = as_f32(map, "map")
    depth = as_f32(depth, "depth")
    b, h, w, _ = map.shape
    if depth.shape[1] != h or depth.shape[2] != w:
        raise ValueError('Height and width of map and depth should be the same')
    rot, trans, f, c = _motion_args(rot, trans, camera, b)
    flow = torch.empty((b, h, w, 2), dtype=torch.float32, device=map.device)
    pmr = torch.empty_like(flow)
    rotc = torch.empty_like(flow)
    check(lib.m4d_reproject_flow(dptr(depth, "depth"), dptr(rot, "rot"), rot.shape[1], dptr(trans, "trans"),
                                 dptr(f), dptr(c), b, h, w, dptr(flow), dptr(pmr), dptr(rotc), stream_ptr()),
          "m4d_reproject_flow")
    return dense_image_warp(map, flow), [pmr, rotc]


def wrap_feature_block(feature_block, opt_flow):
    """utils/depth_operations.py:9-15 (dead code in the reference: it calls the
    TF1-only ``tf.image.resize_bilinear``); provided on the v1 resize kernel."""
    from .network_ops import resize_bilinear_v1
    b, h, w, _ = feature_block.shape
    flow = resize_bilinear_v1(opt_flow, h, w)
    scaled = flow * torch.tensor([float(h), float(w)], device=flow.device)
    return dense_image_warp(feature_block, scaled)


def tile_in_batch(map, nbre_copies):
    """utils/depth_operations.py:217-221: out batch index = copy*b + bi.  A pure
    copy that only exists because the TF graph is unfused; the DSCV kernel never
    materialises it."""
    shape = list(map.shape)
    return map.unsqueeze(0).repeat([nbre_copies] + [1] * len(shape)).reshape([-1] + shape[1:])


_CV_ACCUM = {"fp32_round": 0, "fp16_seq": 1}


def _dscv_forward(c1, c2, disp_prev_t, disp, rot, trans, f, c, r, nbre_cuts, cv_accum, return_index):
    b, h, w, C = c1.shape
    ncp = 2 * r + 1
    cv = torch.empty((b, h, w, nbre_cuts * ncp), dtype=torch.float32, device=c1.device)
    prev_disp = torch.empty((b, h, w, ncp), dtype=torch.float32, device=c1.device)
    idx = torch.empty((b, h, w, ncp, 2), dtype=torch.int32, device=c1.device) if return_index else None
    check(lib.m4d_dscv_fwd(dptr(c1, "c1"), dptr(c2, "c2"), dptr(disp_prev_t), dptr(disp), dptr(rot, "rot"),
                           rot.shape[1], dptr(trans), dptr(f), dptr(c), b, h, w, C, r, nbre_cuts,
                           _CV_ACCUM[cv_accum], dptr(cv), nbre_cuts * ncp, dptr(prev_disp), None, 0, 1.0,
                           dptr(idx, "index", torch.int32), stream_ptr()), "m4d_dscv_fwd")
    return cv, prev_disp, idx


class _DSCV(torch.autograd.Function):
    """get_parallax_sweeping_cv with the gradient tf.GradientTape derives from
    utils/depth_operations.py:224-281 (m4d_dscv_bwd): w.r.t. c1, c2, disp_prev_t and disp."""

    @staticmethod
    def forward(ctx, c1, c2, disp_prev_t, disp, rot, trans, f, c, r, nbre_cuts, cv_accum):
        cv, prev_disp, _ = _dscv_forward(c1, c2, disp_prev_t, disp, rot, trans, f, c, r, nbre_cuts, cv_accum, False)
        ctx.save_for_backward(c1, c2, disp_prev_t, disp, rot, trans, f, c)
        ctx.cfg = (r, nbre_cuts)
        return cv, prev_disp

    @staticmethod
    def backward(ctx, g_cv, g_prev_disp):
        c1, c2, disp_prev_t, disp, rot, trans, f, c = ctx.saved_tensors
        r, nbre_cuts = ctx.cfg
        b, h, w, C = c1.shape
        g_cv = as_f32(g_cv, "g_cv")
        g_prev_disp = as_f32(g_prev_disp, "g_prev_disp") if g_prev_disp is not None else None
        g_c1 = torch.empty_like(c1)
        g_c2 = torch.empty_like(c2)
        g_disp = torch.empty_like(disp)
        g_dpt = torch.empty_like(disp_prev_t) if ctx.needs_input_grad[2] else None
        check(lib.m4d_dscv_bwd(dptr(c1), dptr(c2), dptr(disp_prev_t), dptr(disp), dptr(rot, "rot"), rot.shape[1],
                               dptr(trans), dptr(f), dptr(c), b, h, w, C, r, nbre_cuts, dptr(g_cv, "g_cv"),
                               g_cv.shape[-1], dptr(g_prev_disp, "g_prev_disp"), dptr(g_c1), dptr(g_c2),
                               dptr(g_disp), dptr(g_dpt), stream_ptr()), "m4d_dscv_bwd")
        return g_c1, g_c2, g_dpt, g_disp, None, None, None, None, None, None, None


def get_parallax_sweeping_cv(c1, c2, disp_prev_t, disp, rot, trans, camera, search_range, nbre_cuts=1,
                             cv_accum="fp32_round", return_index=False, out=None):
    """Computes the DSCV as presented in the paper (utils/depth_operations.py:224-281).

    Returns (cv [b,h,w,k*(2r+1)], prev_disp [b,h,w,2r+1]); with ``return_index``
    also the int32 (y0,x0) grid [b,h,w,2r+1,2].  ``cv_accum`` selects how the
    float16 mean of :277 is accumulated (see the oracle's [UNPINNED] note).
    Differentiable w.r.t. c1, c2, disp_prev_t and disp when autograd is recording."""
    c1 = as_f32(c1, "c1")
    c2 = as_f32(c2, "c2")
    b, h, w, C = c1.shape
    if tuple(c2.shape) != (b, h, w, C):
        raise ValueError(f"c1 {tuple(c1.shape)} and c2 {tuple(c2.shape)} must have the same shape")
    if C % nbre_cuts != 0:
        raise ValueError(f"nbre_cuts={nbre_cuts} does not divide the {C} feature channels")
    disp_prev_t = as_f32(disp_prev_t, "disp_prev_t").reshape(b, h, w, 1)
    disp = as_f32(disp, "disp").reshape(b, h, w, 1)
    rot, trans, f, c = _motion_args(rot, trans, camera, b)
    r = int(search_range)
    if torch.is_grad_enabled() and any(t.requires_grad for t in (c1, c2, disp_prev_t, disp)):
        if return_index:
            raise ValueError("return_index is a debugging output; it is not available while recording gradients")
        return _DSCV.apply(c1, c2, disp_prev_t, disp, rot, trans, f, c, r, int(nbre_cuts), cv_accum)
    cv, prev_disp, idx = _dscv_forward(c1, c2, disp_prev_t, disp, rot, trans, f, c, r, int(nbre_cuts), cv_accum,
                                       return_index)
    return (cv, prev_disp, idx) if return_index else (cv, prev_disp)


def _sncv_forward(c1, c2, search_range, dilation_rate, nbre_cuts):
    b, h, w, C = c1.shape
    mo = 2 * search_range + 1
    out = torch.empty((b, h, w, mo * mo * nbre_cuts), dtype=torch.float32, device=c1.device)
    check(lib.m4d_sncv_fwd(dptr(c1, "c1"), dptr(c2, "c2"), b, h, w, C, search_range, dilation_rate,
                           nbre_cuts, dptr(out), mo * mo * nbre_cuts, stream_ptr()), "m4d_sncv_fwd")
    return out


class _SNCV(torch.autograd.Function):
    """cost_volume with its gradient (m4d_sncv_bwd); ``same`` = c1 and c2 are one tensor."""

    @staticmethod
    def forward(ctx, c1, c2, search_range, dilation_rate, nbre_cuts):
        out = _sncv_forward(c1, c2, search_range, dilation_rate, nbre_cuts)
        ctx.save_for_backward(c1, c2, out)
        ctx.cfg = (search_range, dilation_rate, nbre_cuts)
        return out

    @staticmethod
    def backward(ctx, g):
        c1, c2, out = ctx.saved_tensors
        r, d, k = ctx.cfg
        b, h, w, C = c1.shape
        g = as_f32(g, "g")
        g_c1 = torch.empty_like(c1) if ctx.needs_input_grad[0] else None
        g_c2 = torch.empty_like(c2) if ctx.needs_input_grad[1] else None
        check(lib.m4d_sncv_bwd(dptr(c1), dptr(c2), dptr(out), out.shape[-1], dptr(g, "g"), g.shape[-1], b, h, w, C,
                               r, d, k, 0.1, dptr(g_c1), dptr(g_c2), stream_ptr()), "m4d_sncv_bwd")
        return g_c1, g_c2, None, None, None


def cost_volume(c1, c2, search_range, name="cost_volume", dilation_rate=1, nbre_cuts=1):
    """Build cost volume for associating a pixel from Image1 with its
    corresponding pixels in Image2 -- the SNCV (utils/depth_operations.py:284-313).
    Returns [b,h,w,(2r+1)^2*k], channel ((y*(2r+1)+x)*k + kk), leaky_relu(0.1).
    Differentiable w.r.t. c1 and c2 when autograd is recording."""
    c1 = as_f32(c1, "c1")
    c2 = as_f32(c2, "c2")
    b, h, w, C = c1.shape
    if tuple(c2.shape) != (b, h, w, C):
        raise ValueError(f"c1 {tuple(c1.shape)} and c2 {tuple(c2.shape)} must have the same shape")
    if C % nbre_cuts != 0:
        raise ValueError(f"nbre_cuts={nbre_cuts} does not divide the {C} feature channels")
    if torch.is_grad_enabled() and (c1.requires_grad or c2.requires_grad):
        return _SNCV.apply(c1, c2, int(search_range), int(dilation_rate), int(nbre_cuts))
    return _sncv_forward(c1, c2, int(search_range), int(dilation_rate), int(nbre_cuts))
