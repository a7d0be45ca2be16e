"""Winograd kernel 2 vs kernel 4 on the refiner layer shapes that take the Winograd path (levels 1-3).  Run once per
M4D_WINO_VARIANT (2 / 4, with M4D_WINO4_MIN_WG=0 to force kernel 4 everywhere it applies)."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
for (h, w, cin, cout) in [(192, 640, 64, 128), (192, 640, 128, 128), (192, 640, 96, 64), (96, 320, 128, 128), (96, 320, 96, 64),
                          (48, 160, 128, 128), (48, 160, 96, 64), (192, 320, 16, 64)]:
    x = torch.randn(a.batch, h, w, cin, device=dev)
    k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    wu8, cpad = nops.pack_conv_weights_winograd(k.numpy(), chunk=8); wud = torch.from_numpy(wu8).to(dev)
    fn = lambda: nops.conv3x3_wino2_bias_act(x, wud, bias, cout, cpad, 0.1)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / a.iters)
    wg4 = -(-h // 16) * -(-w // 16) * (cpad // 64) * a.batch
    print(f"b={a.batch} {h}x{w} {cin:3d}->{cout:3d}  kernel-4 workgroups {wg4:5d}: {best:8.1f} us", flush=True)
