"""Winograd F(2x2,3x3) MFMA convolution vs the direct MFMA convolution: max difference and time per launch."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
for (h, w, cin, cout) in [(192, 640, 64, 128), (192, 640, 128, 128), (192, 640, 128, 96), (192, 640, 96, 64), (192, 640, 64, 32),
                          (96, 320, 122, 128), (96, 320, 128, 128), (48, 160, 128, 96), (37, 53, 122, 96)]:
    x = torch.randn(a.batch, h, w, cin, device=dev)
    k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    wp, cpad = nops.pack_conv_weights(k.numpy()); wpd = torch.from_numpy(wp).to(dev)
    wu, cpad2 = nops.pack_conv_weights_winograd(k.numpy()); wud = torch.from_numpy(wu).to(dev)
    wu8, _ = nops.pack_conv_weights_winograd(k.numpy(), chunk=8); wu8d = torch.from_numpy(wu8).to(dev)
    v2 = cin % 4 == 0
    ref = nops.conv3x3_bias_act(x, wpd, bias, cout, cpad, 0.1)
    got = nops.conv3x3_wino_bias_act(x, wud, bias, cout, cpad2, 0.1)
    torch.cuda.synchronize()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    res = []
    err2 = float("nan")
    if v2:
        got2 = nops.conv3x3_wino2_bias_act(x, wu8d, bias, cout, cpad2, 0.1)
        torch.cuda.synchronize()
        err2 = (got2 - ref).abs().max().item() / ref.abs().max().item()
    fns = [lambda: nops.conv3x3_bias_act(x, wpd, bias, cout, cpad, 0.1), lambda: nops.conv3x3_wino_bias_act(x, wud, bias, cout, cpad2, 0.1)]
    if v2:
        fns.append(lambda: nops.conv3x3_wino2_bias_act(x, wu8d, bias, cout, cpad2, 0.1))
    for fn in fns:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / a.iters)
    fl = 2 * 9 * cin * cout * h * w * a.batch
    print(f"{h}x{w} b={a.batch} {cin:3d}->{cout:3d}: max|diff|/max|ref| {err:.2e} | direct {res[0]:8.1f} us ({fl/res[0]/1e6:6.1f} TF/s) | winograd {res[1]:8.1f} us ({fl/res[1]/1e6:6.1f} eff. TF/s) | {res[0]/res[1]:.2f}x"
          + (f" | v2 {res[2]:8.1f} us err {err2:.1e} {res[0]/res[2]:.2f}x" if v2 else ""), flush=True)
