"""The bf16-split Winograd kernel (m4d_wino6.hip) against the float32-MFMA Winograd kernels (m4d_wino.hip) on the refiner
layer shapes: time per launch, and the error of both against a float64 convolution (torch on the CPU) in units of the
float32 accumulation scale 2^-24 * sum |x||k| -- the split kernel has to be at least as accurate as the fp32 one."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--check", type=int, default=1)
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


for (h, w, cin, cout) in [(192, 640, 64, 128), (192, 640, 128, 128), (192, 640, 128, 96), (192, 640, 96, 64), (192, 640, 64, 32),
                          (96, 320, 128, 128), (96, 320, 96, 64), (48, 160, 128, 128), (48, 160, 96, 64), (50, 70, 32, 40)]:
    torch.manual_seed(h + cin)
    x = torch.randn(a.batch, h, w, cin, device=dev)
    k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    wu8, cpad8 = nops.pack_conv_weights_winograd(k.numpy(), chunk=8); wud8 = torch.from_numpy(wu8).to(dev)
    wu6, cpad6 = nops.pack_conv_weights_wino6(k.numpy()); wud6 = torch.from_numpy(wu6.view("int16")).to(dev)
    f32 = lambda: nops.conv3x3_wino2_bias_act(x, wud8, bias, cout, cpad8, 0.1)
    f6 = lambda: nops.conv3x3_wino6_bias_act(x, wud6, bias, cout, cpad6, 0.1)
    msg = ""
    if a.check:
        xd = x[:1].cpu().double().permute(0, 3, 1, 2); kd = k.double().permute(3, 2, 0, 1)
        ref = torch.nn.functional.conv2d(xd, kd, bias.cpu().double(), padding=1)
        scale = torch.nn.functional.conv2d(xd.abs(), kd.abs(), None, padding=1) * 2.0 ** -24
        ref = torch.where(ref > 0, ref, ref * 0.1).permute(0, 2, 3, 1)
        scale = scale.permute(0, 2, 3, 1)
        e32 = ((f32()[:1].cpu().double() - ref) / scale).abs()
        e6 = ((f6()[:1].cpu().double() - ref) / scale).abs()
        msg = f"  err/scale fp32-MFMA mean {e32.mean():.3f} max {e32.max():.2f} | bf16x6 mean {e6.mean():.3f} max {e6.max():.2f}"
    t32, t6 = timed(f32, a.iters), timed(f6, a.iters)
    print(f"b={a.batch} {h}x{w} {cin:3d}->{cout:3d}: fp32 {t32:8.1f} us   bf16x6 {t6:8.1f} us  ({t32 / t6:.2f}x){msg}", flush=True)
