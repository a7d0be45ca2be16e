"""Encoder (FeaturePyramid) timing at the bench geometry: the fused head alone and the whole 12-convolution pass on a batch
of 2 frames (the unit bench.py's sequence forward encodes at a time)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4depth_amd as M
from m4depth_amd import synthetic as S, network_ops as nops
dev = torch.device("cuda:0")
model = M.M4Depth(nbre_levels=6); model.load_numpy_weights(S.init_weights(6, seed=42), dev)
img = torch.rand(2, 384, 1280, 3, device=dev)
enc = model.encoder
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
c1, c2, dn = enc.conv_layers_s1[0], enc.conv_layers_s2[0], enc.dn_layers[0]
wp2, cpad2 = c2._packed_weights()
head = lambda: nops.encoder_head(img, c1._hwio_device(), c1.bias, dn.scale, dn.bias, wp2, c2.bias, 16, cpad2, 0.1)
print(f"encoder head (3->16 conv + DINL stats + DINL-fused stride-2 16->16): {timeit(head):.1f} us")
print(f"whole encoder, 2 frames: {timeit(lambda: enc(img)):.1f} us")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    enc(img); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s): enc(img)
print(f"whole encoder, 2 frames, hipGraph replay: {timeit(g.replay):.1f} us")
