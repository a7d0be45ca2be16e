"""CPU ORACLE for the data front end — test infrastructure only, never a product path.

numpy restatement (strict float32, one rounding per operation) of what the reference's
dataloaders do to a decompressed sample (dataloaders/midair.py:33-57, kitti.py:23-52,
tartanair.py:20-47) and of the TF image ops they call.  PARITY UNPINNED (TensorFlow cannot run
here): ``tf.image.resize`` is restated from its documented kernel -- scale = in/out in float32,
bilinear: src = (dst+0.5)*scale-0.5, lower = max(floor,0), upper = min(ceil,in-1),
top = tl+(tr-tl)*xl, out = top+(bot-top)*yl; nearest (half-pixel centres): floor((dst+0.5)*scale).
Only ``tests/`` may import this module.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _axis_bilinear(out_n, in_n):
    scale = F32(in_n) / F32(out_n)
    src = (np.arange(out_n, dtype=F32) + F32(0.5)) * scale - F32(0.5)
    fl = np.floor(src)
    lo = np.clip(fl.astype(np.int64), 0, in_n - 1)
    hi = np.clip(np.ceil(src).astype(np.int64), 0, in_n - 1)
    return lo, hi, (src - fl).astype(F32)


def resize_bilinear(x, out_h, out_w):
    """tf.image.resize(x, [out_h, out_w]) (bilinear, antialias=False) on [...,H,W,C] float32."""
    x = np.asarray(x, F32)
    ih, iw = x.shape[-3], x.shape[-2]
    ylo, yhi, yl = _axis_bilinear(out_h, ih)
    xlo, xhi, xl = _axis_bilinear(out_w, iw)
    xl = xl.reshape(1, out_w, 1)
    yl = yl.reshape(out_h, 1, 1)
    tl = x[..., ylo, :, :][..., :, xlo, :]
    tr = x[..., ylo, :, :][..., :, xhi, :]
    bl = x[..., yhi, :, :][..., :, xlo, :]
    br = x[..., yhi, :, :][..., :, xhi, :]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return (top + (bot - top) * yl).astype(F32)


def resize_nearest(x, out_h, out_w):
    """tf.image.resize(x, size, method='nearest') on [...,H,W,C]."""
    x = np.asarray(x)
    ih, iw = x.shape[-3], x.shape[-2]
    yi = np.minimum(np.floor((np.arange(out_h, dtype=F32) + F32(0.5)) * (F32(ih) / F32(out_h))).astype(np.int64), ih - 1)
    xi = np.minimum(np.floor((np.arange(out_w, dtype=F32) + F32(0.5)) * (F32(iw) / F32(out_w))).astype(np.int64), iw - 1)
    return x[..., yi, :, :][..., :, xi, :]


def decode_rgb(image_u8, out_h, out_w):
    """cast(image)/255 then resize (midair.py:35-45)."""
    return resize_bilinear(np.asarray(image_u8).astype(F32) / F32(255.0), out_h, out_w)


def decode_depth_midair(png_u16, out_h, out_w):
    """midair.py:49-55: uint16 -> bitcast float16 -> 512/x -> bilinear resize.  [H,W] -> [h,w,1]."""
    half = np.asarray(png_u16, np.uint16).view(np.float16)
    with np.errstate(divide="ignore"):
        depth = (F32(512.0) / half.astype(F32)).astype(F32)
    return resize_bilinear(depth[..., None], out_h, out_w)


def kitti_eval_crop(out_h, out_w):
    """kitti.py:14-20: (y0, y1, x0, x1) of the Garg/Eigen crop."""
    crop = np.array([0.40810811 * out_h, 0.99189189 * out_h, 0.03594771 * out_w, 0.96405229 * out_w]).astype(np.int32)
    return tuple(int(v) for v in crop)


def decode_depth_kitti(png_u16, out_h, out_w, eval_crop):
    """kitti.py:43-50: uint16/256, nearest resize, (eval) crop mask."""
    depth = (np.asarray(png_u16, np.uint16).astype(F32) / F32(256.0))[..., None]
    out = resize_nearest(depth, out_h, out_w).astype(F32)
    if eval_crop:
        y0, y1, x0, x1 = kitti_eval_crop(out_h, out_w)
        mask = np.zeros([out_h, out_w, 1], F32)
        mask[y0:y1, x0:x1] = 1
        out = out * mask
    return out


def decode_depth_tartanair(raw_f32, rgb_resized, out_h, out_w):
    """tartanair.py:37-45: float32 map, nearest resize, zero where the resized colour is black."""
    out = resize_nearest(np.asarray(raw_f32, F32)[..., None], out_h, out_w).astype(F32)
    rgb = np.asarray(rgb_resized, F32)
    nrm = np.sqrt((rgb[..., 0:1] * rgb[..., 0:1] + rgb[..., 1:2] * rgb[..., 1:2]) + rgb[..., 2:3] * rgb[..., 2:3])
    return out * (nrm > 0).astype(F32)


def flip_sequence(sample, vertical, horizontal, h, w):
    """_augmentation_step_flip (generic.py:208-259) with the two coin flips given."""
    col, dep = sample["RGB_im"], sample["depth"]
    rot, trans = sample["rot"].copy(), sample["trans"].copy()
    c = np.array(sample["camera"]["c"], F32)
    if vertical:
        col, dep = col[:, ::-1], dep[:, ::-1]
        rot = rot * np.array([[1., -1., 1., -1.]], F32)
        trans = trans * np.array([[1., -1., 1.]], F32)
        c = np.array([c[0], h - c[1]], F32)
    if horizontal:
        col, dep = col[:, :, ::-1], dep[:, :, ::-1]
        rot = rot * np.array([[1., 1., -1., -1.]], F32)
        trans = trans * np.array([[-1., 1., 1.]], F32)
        c = np.array([w - c[0], c[1]], F32)
    return {"RGB_im": col, "depth": dep, "rot": rot, "trans": trans, "camera": {"f": sample["camera"]["f"], "c": c}}
