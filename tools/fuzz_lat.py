"""Random-shape check of m4d_conv3x3_lat / m4d_conv3x3s_lat (csrc/m4d_convlat.hip): random maps, channel counts (Cin % 4 == 0, any
Cout), strides, batch, (mt, kw, s_out) and 1-4 input slabs against the float64 convolution (tolerance of the tests); equal (kw, s_out)
give equal bits whatever mt; a partial-slab input gives the bits of its finished tensor; repeated launches identical."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=80); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
bad = 0
for case in range(a.cases):
    b = int(rng.integers(1, 4)); h = int(rng.integers(2, 40)); w = int(rng.integers(2, 50))
    cin = 4 * int(rng.integers(4, 60)); cout = int(rng.choice([4, 5, 16, 24, 32, 33, 40, 64, 96, 100, 128, 192]))
    stride = int(rng.choice([1, 1, 2])); slope = float(rng.choice([0.1, 1.0]))
    nch = -(-cin // 16)
    kw = int(rng.choice([1, 2, 4])); s_out = int(rng.integers(1, 5))
    while s_out > nch or (s_out - 1) * (-(-nch // s_out)) >= nch:
        s_out -= 1
    mts = [m for m in (1, 2, 4, 8) if 2 * (1 if m == 8 else kw) * nops._lat_halo_pixels(m, stride) * 96 <= 160 * 1024]
    s_in = int(rng.integers(1, 5))
    slabs = torch.from_numpy(rng.standard_normal([s_in, b, h, w, cin]).astype(np.float32)).to(dev)
    xb = torch.from_numpy((0.1 * rng.standard_normal([cin])).astype(np.float32)).to(dev)
    x_in = slabs[0] if s_in == 1 else nops.PartialAct(slabs, xb, 0.1)
    x_dense = slabs[0] if s_in == 1 else x_in.dense()
    k = (rng.standard_normal([3, 3, cin, cout]) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    bias = torch.from_numpy((0.1 * rng.standard_normal([cout])).astype(np.float32)).to(dev)
    wd = torch.from_numpy(nops.pack_conv_weights_lat(k).view(np.int16)).to(dev)
    oh, ow = -(-h // stride), -(-w // stride)
    ph, pw = max((oh - 1) * stride + 3 - h, 0), max((ow - 1) * stride + 3 - w, 0)
    xp = torch.nn.functional.pad(x_dense.double().permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    ref = torch.nn.functional.conv2d(xp, torch.from_numpy(k).to(dev).double().permute(3, 2, 0, 1), bias.double(), stride)
    ref = torch.nn.functional.leaky_relu(ref, slope).permute(0, 2, 3, 1)
    ok, first, err = True, {}, 0.0
    for mt in mts:
        kk = 1 if mt == 8 else kw
        out = nops.conv3x3_lat(x_in, wd, bias, cout, slope, config=(mt, kk, s_out), stride=stride)
        again = nops.conv3x3_lat(x_dense, wd, bias, cout, slope, config=(mt, kk, s_out), stride=stride)
        if s_out > 1:
            if cout % 4:
                continue
            out, again = out.dense(), again.dense()
        ok = ok and torch.equal(out, again)                       # slabs finished while staging == finished first; deterministic
        err = max(err, float((out.double() - ref).abs().max()) / max(1.0, float(ref.abs().max())))
        key = (kk, s_out)
        if key in first:
            ok = ok and torch.equal(out, first[key])
        else:
            first[key] = out
    ok = ok and err < 1e-5
    bad += 0 if ok else 1
    print(f"case {case:3d}: b={b} {h}x{w} {cin}->{cout} stride {stride} s_in {s_in} (kw {kw}, s_out {s_out}) mts {mts}: max rel err {err:.2e}  "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{bad} cases failed")
sys.exit(1 if bad else 0)
