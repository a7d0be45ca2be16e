"""Encoder level 0 (m4d_enc_level0_fwd: RGB -> conv 3->16 -> DomainNormalization -> leaky_relu -> conv 16->16 stride 2, five
kernels, no full-resolution intermediate) on a batch of frames, graph-replayed: us per call; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4depth_amd as M
from m4depth_amd import synthetic as S, network_ops as nops
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=2); ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = M.M4Depth(nbre_levels=6); model.load_numpy_weights(S.init_weights(6, seed=42), dev)
img = torch.rand(a.frames, 384, 1280, 3, device=dev)
enc = model.encoder
c1, c2, dn = enc.conv_layers_s1[0], enc.conv_layers_s2[0], enc.dn_layers[0]
fn = lambda: nops.encoder_level0(img, c1._hwio_device(), c1.bias, dn.scale, dn.bias, c2._hwio_device(), c2.bias, 0.1)
for _ in range(3): fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(a.iters): fn()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"encoder level 0, {a.frames} frames of 384x1280: {e0.elapsed_time(e1) * 1e3 / a.iters:.1f} us per call")
