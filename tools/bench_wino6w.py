"""The wide (m4d_wino6w.hip) and half-tile (m4d_wino6h.hip) bf16-split Winograd kernels against the 16x16 x 64-cout kernel
(m4d_wino6.hip): bit equality and time per launch on the refiner / encoder layer shapes they serve."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
from m4depth_amd._lib import require_experiments
require_experiments("tools/bench_wino6w.py")
from m4depth_amd._lib import lib
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


for (h, w, cin, cout) in [(192, 640, 64, 128), (192, 640, 128, 128), (192, 640, 128, 96), (96, 320, 128, 128), (96, 320, 128, 96),
                          (96, 320, 96, 64), (96, 320, 32, 64), (48, 160, 128, 128), (48, 160, 128, 96), (48, 160, 96, 64), (48, 160, 64, 96),
                          (24, 80, 128, 128), (50, 70, 32, 100), (33, 47, 48, 128)]:
    torch.manual_seed(h + cin)
    x = torch.randn(a.batch, h, w, cin, device=dev)
    k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    wu6, cpad6 = nops.pack_conv_weights_wino6(k.numpy()); wud6 = torch.from_numpy(wu6.view("int16")).to(dev)
    f6 = lambda: nops.conv3x3_wino6_bias_act(x, wud6, bias, cout, cpad6, 0.1)
    lib.m4d_wino6_set_variant(1); ref = f6(); t_n = timed(f6, a.iters)
    if cout > 64 and cout % 4 == 0:
        lib.m4d_wino6_set_variant(2); got = f6(); t_w = timed(f6, a.iters)
    else:
        got, t_w = ref, float("nan")
    lib.m4d_wino6_set_variant(3); goth = f6(); t_h = timed(f6, a.iters)
    lib.m4d_wino6_set_variant(4); gotb = f6(); t_b = timed(f6, a.iters)
    bad_b = sum(int(not torch.equal(f6(), ref)) for _ in range(30))          # a race in the barrier scheme would show up here
    lib.m4d_wino6_set_variant(0)
    wgs = a.batch * -(-h // 16) * -(-w // 16) * -(-cout // 64)
    print(f"b={a.batch} {h}x{w} {cin:3d}->{cout:3d} ({wgs:4d} wg): 16x16x64 {t_n:7.1f} us   wide {t_w:7.1f} us ({t_n / t_w:.2f}x, same bits {torch.equal(ref, got)})"
          f"   half-tile {t_h:7.1f} us ({t_n / t_h:.2f}x, same bits {torch.equal(ref, goth)})"
          f"   one barrier per two positions {t_b:7.1f} us ({t_n / t_b:.2f}x, same bits {torch.equal(ref, gotb)}, {bad_b} of 30 repeats differ)", flush=True)
