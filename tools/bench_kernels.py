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
os.environ["M4D_FUSED_FRONT"] = "0"                          # the model step below must leave the separate kernels' inputs behind
import m4depth_amd as M                                     # noqa: E402
from m4depth_amd import synthetic as S                      # noqa: E402
from m4depth_amd._lib import lib, dptr, stream_ptr          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--which", default="dscv,sncv")
    ap.add_argument("--ablate", type=int, default=0)
    ap.add_argument("--smooth", action="store_true", help="replace the parallax maps by smooth fields (coherent gathers)")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--dscv-range", type=int, default=4)
    ap.add_argument("--sncv-range", type=int, default=3, help="with --height 768 --width 2560 --dscv-range 6 --sncv-range 6: BASELINE configs[4]")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H, Wd, L = args.height, args.width, 6
    RD, RS = args.dscv_range, args.sncv_range
    NCP, MO2 = 2 * RD + 1, (2 * RS + 1) ** 2
    model = M.M4Depth(nbre_levels=L, dscv_range=RD, sncv_range=RS)
    model.load_numpy_weights(S.init_weights(L, seed=42, dscv_range=RD, sncv_range=RS), dev)
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
    bytes_ = {"dscv": 4 * px * (2 * C + 2 + NCP * k + 1), "sncv": 4 * px * (C + MO2 * k)}
    fin = f_input.data_ptr()

    def run_dscv():
        return lib.m4d_dscv_fwd(dptr(c1), dptr(c2), dptr(dpt), dptr(disp), dptr(rot), rot.shape[1], dptr(tr), dptr(cf),
                                dptr(cc), b, h, w, C, RD, k, 0, ctypes.c_void_p(fin), F_in, None,
                                ctypes.c_void_p(fin + 4 * (F_in - 1)), F_in, 0.25, None, stream_ptr())

    def run_sncv():
        return lib.m4d_sncv_fwd(dptr(c1), dptr(c1), b, h, w, C, RS, 1, k, ctypes.c_void_p(fin + 4 * (NCP * k + 5)), F_in,
                                stream_ptr())

    raw_f, prev_l, depth_t = lvl.last_front_inputs
    F_st = (F_in + 7) // 8 * 8
    f_front = torch.zeros((b, h, w, F_st), device=dev)
    norm_out = torch.empty_like(c1)
    bytes_["front"] = int(px * (4 * (3 * C + 1 + F_in) + 5))

    def run_front():
        pp = prev_l["parallax"] if prev_l is not None else None
        po = prev_l["other"] if prev_l is not None else None
        ph, pw = (pp.shape[1], pp.shape[2]) if pp is not None else (0, 0)
        return lib.m4d_level_front_r(dptr(raw_f), dptr(norm_out), dptr(c2), dptr(depth_t), dptr(pp), dptr(po), ph, pw,
                                     dptr(rot), rot.shape[1], dptr(tr), dptr(cf), dptr(cc), b, h, w, C, k, RD, RS, 0,
                                     dptr(f_front), F_st, 0.25, stream_ptr())

    lib.m4d_dscv_set_ablation(args.ablate)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    variants = [("dscv[wave]", run_dscv, 1), ("dscv[lds-window]", run_dscv, 2), ("dscv[lds-hyp]", run_dscv, 3), ("dscv[lds-hyp9]", run_dscv, 4),
                ("sncv", run_sncv, None), ("front", run_front, None)]
    bytes_["dscv[wave]"] = bytes_["dscv[lds-window]"] = bytes_["dscv[lds-hyp]"] = bytes_["dscv[lds-hyp9]"] = bytes_["dscv"]
    for name, fn, variant in variants:
        if name.split("[")[0] not in args.which.split(",") and name not in args.which.split(","):
            continue
        if variant is not None:
            lib.m4d_dscv_set_variant(variant)
            lib.m4d_dscv_set_fallback_counter(ctypes.c_void_p(counter.data_ptr()))
            counter.zero_()
            fn()
            torch.cuda.synchronize()
            tiles = -(-w // 32) * -(-h // 8) * b
            print(f"  {name}: fallback workgroups {int(counter.item())} of ~{tiles}", flush=True)
            lib.m4d_dscv_set_fallback_counter(None)
            if variant in (3, 4):
                stamps = torch.zeros((tiles * 4, 6), dtype=torch.int64, device=dev)
                lib.m4d_dscv_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
                fn()
                torch.cuda.synchronize()
                lib.m4d_dscv_set_stamps(None)
                st = stamps[stamps[:, 0] != 0].double()
                if len(st):
                    d = [float((st[:, i + 1] - st[:, i]).mean()) for i in range(3)]
                    print(f"  {name}: cycles/workgroup box {d[0]:.0f}  stage {d[1]:.0f}  gather+compute {d[2]:.0f}  | window px mean "
                          f"{float(st[:, 4].mean()):.0f} max {float(st[:, 4].max()):.0f}, width mean {float(st[:, 5].mean()):.1f}; "
                          f"kernel span {(float(st[:, 3].max()) - float(st[:, 0].min())) / 1e3:.0f} kcycles over {len(st)} workgroups", flush=True)
        if name == "front":
            tiles = max(4096, b * h * w // 32 + 64)       # 8 stamps per WORKGROUP: the smallest front tile is 8x4 pixels (a fixed 4096 rows
                                                          # let batch-32 runs write past the buffer: a GPU memory fault under rocprofv3)
            stamps = torch.zeros((tiles, 8), dtype=torch.int64, device=dev)
            lib.m4d_front_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
            fn()
            torch.cuda.synchronize()
            lib.m4d_front_set_stamps(None)
            st = stamps[stamps[:, 0] != 0].double()
            if len(st):
                names = ["stage+upsample", "normalise", "state store+SNCV", "c1+stage writes", "DSCV", "row stores"]
                d = [float((st[:, i + 1] - st[:, i]).mean()) for i in range(6)]
                print("  front phases (cycles/workgroup, mean over %d): " % len(st) + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, d))
                      + f" | total {float((st[:, 6] - st[:, 0]).mean()):.0f}; kernel span {(float(st[:, 6].max()) - float(st[:, 0].min())) / 1e3:.1f} kcycles", flush=True)
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
