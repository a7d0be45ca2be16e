"""Per-layer time of one encoder pass (FeaturePyramid, b frames of 384x1280) in the product's dispatch: level 0 as one call,
then conv stride 1 / conv stride 2 per level, and the whole pass."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from m4depth_amd import network as net, synthetic as S
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=2); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
L = 6
model = net.M4Depth(nbre_levels=L)
model.load_numpy_weights(S.init_weights(L, seed=42), dev)
enc = model.encoder
img = torch.rand(a.batch, 384, 1280, 3, device=dev)


def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / a.iters)
    return best


outs = enc(img)
print(f"whole encoder pass, b={a.batch}: {timed(lambda: enc(img)):.1f} us")
x = outs[0]
for i in range(1, L):
    c1, c2 = enc.conv_layers_s1[i], enc.conv_layers_s2[i]
    t1 = timed(lambda: c1(x, slope=0.1))
    y = c1(x, slope=0.1)
    t2 = timed(lambda: c2(y, slope=0.1))
    k1 = net._use_winograd(1, x.shape[1], x.shape[2], x.shape[3], c1.out_channels, 1)
    print(f"level {i}: s1 {x.shape[3]:3d}->{c1.out_channels:3d} at {x.shape[1]}x{x.shape[2]} (kernel kind {k1}): {t1:6.1f} us;  "
          f"s2 {c1.out_channels}->{c2.out_channels} -> {y.shape[1] // 2}x{y.shape[2] // 2}: {t2:6.1f} us")
    x = c2(y, slope=0.1)
