"""Encoder level 0: the two-call head (conv 3->16 + statistics, stride-2 conv with the normalisation fused in: writes the
[b,H,W,16] map) against m4d_enc_level0_fwd (three recomputing passes, no intermediate).  us per call, b frames of 384x1280."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=2); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
h, w = 384, 1280
img = torch.rand(a.batch, h, w, 3, device=dev)
k1 = torch.randn(3, 3, 3, 16) * (2.0 / 27) ** 0.5
k2 = torch.randn(3, 3, 16, 16) * (2.0 / 144) ** 0.5
b1, b2 = torch.randn(16, device=dev) * 0.1, torch.randn(16, device=dev) * 0.1
sc, bs = torch.ones(16, device=dev), torch.zeros(16, device=dev)
wp2, cpad2 = nops.pack_conv_weights(k2.numpy()); wp2 = torch.from_numpy(wp2).to(dev)
k1d, k2d = k1.to(dev), k2.to(dev)
old = lambda: nops.encoder_head(img, k1d.reshape(27, 16), b1, sc, bs, wp2, b2, 16, cpad2, 0.1)
new = lambda: nops.encoder_level0(img, k1d, b1, sc, bs, k2d, b2, 0.1)
print("max |difference| between the two paths:", float((old() - new()).abs().max()))
for name, fn in (("two-call head", old), ("m4d_enc_level0_fwd", new)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / a.iters)
    print(f"{name:22s} b={a.batch}: {best:7.1f} us")
