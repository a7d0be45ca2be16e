"""m4d_wino6p.hip (persistent workgroups, kernel=2) against m4d_wino6.hip (one workgroup per unit, kernel=1) on the layer shapes
the bf16-split Winograd kernel serves: time per launch alone, bit equality."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")


def timed(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


for (h, w, cin, cout) in [(192, 640, 64, 128), (192, 640, 128, 128), (192, 640, 128, 96), (192, 640, 96, 64),
                          (96, 320, 128, 128), (96, 320, 128, 96), (96, 320, 96, 64), (48, 160, 128, 128), (48, 160, 96, 64),
                          (192, 640, 32, 64), (96, 320, 64, 64)]:
    torch.manual_seed(h + cin)
    x = torch.randn(a.batch, h, w, cin, device=dev)
    k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    wu6, cpad6 = nops.pack_conv_weights_wino6(k.numpy()); wud6 = torch.from_numpy(wu6.view("int16")).to(dev)
    f1 = lambda: nops.conv3x3_wino6_bias_act(x, wud6, bias, cout, cpad6, 0.1, kernel=1)
    f2 = lambda: nops.conv3x3_wino6_bias_act(x, wud6, bias, cout, cpad6, 0.1, kernel=2)
    same = torch.equal(f1(), f2())
    t1, t2 = timed(f1, a.iters), timed(f2, a.iters)
    units = a.batch * (-(-h // 16)) * (-(-w // 16)) * (-(-cout // 64))
    print(f"b={a.batch} {h}x{w} {cin:3d}->{cout:3d} ({units:5d} units): one-unit {t1:8.1f} us   persistent {t2:8.1f} us  ({t1 / t2:.3f}x)  bits equal: {same}", flush=True)
