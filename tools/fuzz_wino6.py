"""Random-shape check of the bf16-split Winograd kernels: one workgroup per unit (kernel=1) against persistent workgroups
(kernel=2, Cin >= 32) bit for bit, both against the float64 convolution within the float32 tolerance of the tests, repeated
launches identical.  Shapes: Cin a multiple of 16, any Cout (full units, half units, partly filled N-tiles, scalar stores), ragged
tiles, 1-3 cout groups, 1-15 K chunks."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
bad = 0
for case in range(a.cases):
    b = int(rng.integers(1, 5)); h = int(rng.integers(5, 120)); w = int(rng.integers(5, 150))
    cin = 16 * int(rng.integers(1, 16)); cout = int(rng.choice([8, 24, 32, 33, 40, 64, 66, 96, 100, 128, 160, 192]))
    slope = float(rng.choice([0.1, 1.0]))
    x = torch.from_numpy(rng.standard_normal([b, h, w, cin]).astype(np.float32)).to(dev)
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    bias = torch.from_numpy((0.1 * rng.standard_normal([cout])).astype(np.float32)).to(dev)
    wu6, cpad = nops.pack_conv_weights_wino6(k); wud = torch.from_numpy(wu6.view("int16")).to(dev)
    one = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, slope, kernel=1)
    again = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, slope, kernel=1)
    ok = torch.equal(one, again)
    if cin >= 32:
        per = nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, slope, kernel=2)
        ok = ok and torch.equal(one, per)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), torch.from_numpy(k).to(dev).double().permute(3, 2, 0, 1), bias.double(), padding=1)
    ref = torch.nn.functional.leaky_relu(ref, slope).permute(0, 2, 3, 1)
    err = float((one.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    ok = ok and err < 1e-5
    if not ok:
        bad += 1
    print(f"case {case:3d}: b={b} {h}x{w} {cin}->{cout} slope {slope}: max rel err {err:.2e}  {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{bad} / {a.cases} cases failed")
sys.exit(1 if bad else 0)
