"""Image warping using per-pixel flow vectors -- MI355X native.

Same public surface as the reference module ``utils/dense_image_warp.py``:
``dense_image_warp``, ``_interpolate_bilinear``, ``back_project``,
``back_project_grad`` and the ``use_cuda_backproject`` switch.  Tensors are
``torch.Tensor`` on the ROCm device, NHWC float32.  Every function dispatches to
a HIP kernel of libm4depth_hip.so; there is no PyTorch/CPU implementation here.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import lib, dptr, stream_ptr, check, as_f32

# The reference flips this to True when utils/special_ops/backproject.so loads
# (utils/dense_image_warp.py:38-58) and then routes dense_image_warp through the
# CUDA op (clipped query, weights form).  Default False = the TF-CPU numerics
# (_interpolate_bilinear), which is the parity target; set True for the op route.
use_cuda_backproject = False


def _dims6(coords, inputs):
    b, h, w, s, f = coords.shape[:5]
    c = inputs.shape[4]
    return (ctypes.c_int * 6)(b, h, w, s, f, c), (b, h, w, s, f, c)


class _BackProject(torch.autograd.Function):
    """BackProject with the gradient registered at utils/dense_image_warp.py:46-52."""

    @staticmethod
    def forward(ctx, inputs, coords):
        inputs = as_f32(inputs, "inputs")
        coords = as_f32(coords, "coords")
        if inputs.dim() != 5 or coords.dim() != 6 or coords.shape[-1] != 2:
            raise ValueError("back_project expects inputs [B,H,W,F,C] and coords [B,H,W,S,F,2]")
        if tuple(inputs.shape[:3]) != tuple(coords.shape[:3]) or inputs.shape[3] != coords.shape[4]:
            raise ValueError(f"back_project: inputs {tuple(inputs.shape)} and coords {tuple(coords.shape)} disagree")
        dims, shape = _dims6(coords, inputs)
        out = torch.empty(shape, dtype=torch.float32, device=inputs.device)
        check(lib.m4d_backproject_fwd(dptr(inputs, "inputs"), dptr(coords, "coords"), dims, dptr(out), stream_ptr()),
              "m4d_backproject_fwd")
        ctx.save_for_backward(inputs, coords)
        return out

    @staticmethod
    def backward(ctx, grad):
        inputs, coords = ctx.saved_tensors
        gi, gc = back_project_grad(inputs, coords, grad)
        return gi, gc


def back_project(inputs, coords):
    """BackProject op (cuda_backproject/backproject_op.cc:32-35, kernel
    backproject_op_gpu.cu.cc:19-79): inputs [B,H,W,F,C], coords [B,H,W,S,F,2]
    as (x, y) -> [B,H,W,S,F,C]; coordinates outside the image give zeros."""
    return _BackProject.apply(inputs, coords)


def back_project_grad(inputs, coords, grad):
    """BackProjectGrad op (backproject_op.cc:37-42, kernel
    backproject_op_gpu.cu.cc:108-197) -> (inputs_grad, coords_grad)."""
    inputs = as_f32(inputs, "inputs")
    coords = as_f32(coords, "coords")
    grad = as_f32(grad, "grad")
    dims, shape = _dims6(coords, inputs)
    if tuple(grad.shape) != shape:
        raise ValueError(f"back_project_grad: grad {tuple(grad.shape)} != {shape}")
    gi = torch.empty_like(inputs)
    gc = torch.empty_like(coords)
    check(lib.m4d_backproject_bwd(dptr(grad, "grad"), dptr(inputs, "inputs"), dptr(coords, "coords"), dims,
                                  dptr(gi), dptr(gc), stream_ptr()), "m4d_backproject_bwd")
    return gi, gc


def _interpolate_bilinear(grid, query_points, name='interpolate_bilinear', indexing='ij', return_index=False):
    """utils/dense_image_warp.py:61-192.  grid [B,H,W,C], query_points [B,N,2]
    -> [B,N,C].  ``return_index`` additionally returns the int32 (y0, x0)
    floors [B,N,2] -- the bit-exact index grid of the parity contract."""
    if indexing != 'ij' and indexing != 'xy':
        raise ValueError('Indexing mode must be \'ij\' or \'xy\'')
    grid = as_f32(grid, "grid")
    if grid.dim() != 4:
        raise ValueError('Grid must be 4 dimensional. Received size: ' + str(tuple(grid.shape)))
    query_points = as_f32(query_points, "query_points")
    if query_points.dim() != 3 or query_points.shape[2] != 2:
        raise ValueError('Query points must be 3 dimensional and size 2 in dim 2.')
    if indexing == 'xy':
        query_points = query_points.flip(-1).contiguous()
    B, H, W, C = grid.shape
    if H < 2 or W < 2:
        raise ValueError('Grid height and width must be at least 2.')
    N = query_points.shape[1]
    out = torch.empty((B, N, C), dtype=torch.float32, device=grid.device)
    idx = torch.empty((B, N, 2), dtype=torch.int32, device=grid.device) if return_index else None
    check(lib.m4d_interpolate_bilinear(dptr(grid, "grid"), dptr(query_points, "query_points"), B, H, W, C, N,
                                       dptr(out), dptr(idx, "index", torch.int32), stream_ptr()),
          "m4d_interpolate_bilinear")
    return (out, idx) if return_index else out


def dense_image_warp(image, flow, name='dense_image_warp', return_index=False):
    """utils/dense_image_warp.py:195-268.  image [B,H,W,C] (or [H,W,C]), flow
    [B,H,W,2] in (row, col) order; output[b,j,i] samples image at
    (j + flow[...,0], i + flow[...,1]) (query = grid + flow, :244) with
    border-replicating bilinear interpolation."""
    image = as_f32(image, "image")
    squeeze = image.dim() == 3
    if squeeze:
        image = image.unsqueeze(0)
    if image.dim() != 4:
        raise ValueError("image must be [batch, height, width, channels]")
    B, H, W, C = image.shape
    flow = as_f32(flow, "flow").reshape(B, H, W, 2)
    if H < 2 or W < 2:
        raise ValueError("dense_image_warp needs height >= 2 and width >= 2")
    if use_cuda_backproject and not return_index:
        # the op route of :246-253: clip, reverse to (x, y), S = F = 1
        gy, gx = torch.meshgrid(torch.arange(H, device=image.device, dtype=torch.float32),
                                torch.arange(W, device=image.device, dtype=torch.float32), indexing="ij")
        q = torch.stack([gy, gx], dim=-1).unsqueeze(0) + flow
        lo = torch.zeros(2, device=image.device)
        hi = torch.tensor([float(H - 1), float(W - 1)], device=image.device)
        q = torch.minimum(torch.maximum(q, lo), hi)
        coords = q.flip(-1).reshape(B, H, W, 1, 1, 2).contiguous()
        out = back_project(image.reshape(B, H, W, 1, C), coords).reshape(B, H, W, C)
        return out[0] if squeeze else out
    out = torch.empty_like(image)
    idx = torch.empty((B, H, W, 2), dtype=torch.int32, device=image.device) if return_index else None
    check(lib.m4d_dense_image_warp(dptr(image, "image"), dptr(flow, "flow"), B, H, W, C, dptr(out),
                                   dptr(idx, "index", torch.int32), stream_ptr()), "m4d_dense_image_warp")
    if squeeze:
        out = out[0]
    return (out, idx) if return_index else out
