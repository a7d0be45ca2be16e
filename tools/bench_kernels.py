"""Stand-alone timing of the hand-written cost-volume kernels on the tensors of a real
model step (level ``--level`` of the 384x1280 pyramid, batch ``--batch``), for quick
iteration and for rocprofv3 --pmc passes (many identical launches, nothing else)."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4depth_amd as M                                     # noqa: E402
from m4depth_amd import synthetic as S                      # noqa: E402
from m4depth_amd._lib import lib, dptr, stream_ptr          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--which", default="dscv,sncv")
    ap.add_argument("--smooth", action="store_true", help="replace the parallax maps by smooth fields (coherent gathers)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H, Wd, L = 384, 1280, 6
    model = M.M4Depth(nbre_levels=L)
    model.load_numpy_weights(S.init_weights(L, seed=42), dev)
    samples, cam = S.make_sequence(min(args.batch, 2), 3, H, Wd, seed=1235)
    reps = -(-args.batch // min(args.batch, 2))

    def dv(x):
        if isinstance(x, dict):
            return {k: dv(v) for k, v in x.items()}
        if isinstance(x, list):
            return [dv(v) for v in x]
        if x.dtype == np.bool_:
            return torch.from_numpy(np.concatenate([x] * reps)[:args.batch])
        return torch.from_numpy(np.concatenate([x] * reps)[:args.batch]).to(dev)

    model([dv(samples), dv(cam)])
    lvl = model.d_estimator.levels[args.level - 1]
    c1, c2, dpt, disp, rot, tr, cf, cc = lvl.last_cv_inputs
    b, h, w, C = c1.shape
    k = lvl.nbre_cuts
    if args.smooth:
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h, device=dev), torch.linspace(0, 1, w, device=dev), indexing="ij")
        disp = (2.0 + 3.0 * torch.sin(3 * xx) * torch.cos(2 * yy)).reshape(1, h, w, 1).repeat(b, 1, 1, 1).contiguous()
        dpt = disp.clone()
    F_in = lvl.f_in
    f_input = torch.empty((b, h, w, F_in), device=dev)
    px = b * h * w
    bytes_ = {"dscv": 4 * px * (2 * C + 2 + 9 * k + 1), "sncv": 4 * px * (C + 49 * k)}
    fin = f_input.data_ptr()

    def run_dscv():
        return lib.m4d_dscv_fwd(dptr(c1), dptr(c2), dptr(dpt), dptr(disp), dptr(rot), rot.shape[1], dptr(tr), dptr(cf),
                                dptr(cc), b, h, w, C, 4, k, 0, ctypes.c_void_p(fin), F_in, None,
                                ctypes.c_void_p(fin + 4 * (F_in - 1)), F_in, 0.25, None, stream_ptr())

    def run_sncv():
        return lib.m4d_sncv_fwd(dptr(c1), dptr(c1), b, h, w, C, 3, 1, k, ctypes.c_void_p(fin + 4 * (9 * k + 5)), F_in,
                                stream_ptr())

    for name, fn in (("dscv", run_dscv), ("sncv", run_sncv)):
        if name not in args.which.split(","):
            continue
        for _ in range(3):
            assert fn() == 0
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        print(f"{name} level {args.level} b={b} {h}x{w} C={C} k={k}: {us:9.2f} us/launch  "
              f"{bytes_[name] / us / 1e3:8.1f} GB/s algorithmic  ({bytes_[name] / 1e6:.1f} MB)"
              + ("  [smooth parallax]" if args.smooth else ""), flush=True)


if __name__ == "__main__":
    main()
