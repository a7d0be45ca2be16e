"""Winograd kernel 2 (M4D_WINO_VARIANT=2) vs kernel 4 (default where the grid is large enough; M4D_WINO4_MIN_WG=0 forces
it wherever it applies): run once per variant with --save, then --compare: the outputs must be bit-identical (same
arithmetic, same order).  The inputs are saved next to the outputs, so the test can also compare both with the CPU oracle.
Used by tests/test_gpu_ops.py::test_winograd_kernel_4_is_bitwise_kernel_2."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser()
ap.add_argument("--save"); ap.add_argument("--compare", nargs=2)
a = ap.parse_args()
if a.compare:
    x, y = torch.load(a.compare[0])["out"], torch.load(a.compare[1])["out"]
    for k in x:
        same = torch.equal(x[k].view(torch.int32), y[k].view(torch.int32))
        print(k, "bit-identical" if same else f"DIFFERENT max {(x[k] - y[k]).abs().max().item():.3e}")
    sys.exit(0)
dev = torch.device("cuda:0")
out = {"in": {}, "out": {}}
g = torch.Generator().manual_seed(3)
for (b, h, w, cin, cout) in [(1, 192, 640, 128, 128), (2, 96, 320, 124, 64), (1, 37, 53, 36, 40), (1, 50, 70, 44, 128), (3, 16, 16, 32, 64),
                             (1, 100, 130, 64, 120)]:
    x = torch.randn(b, h, w, cin, generator=g).to(dev)
    k = torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    wu8, cpad = nops.pack_conv_weights_winograd(k.numpy(), chunk=8)
    wud = torch.from_numpy(wu8).to(dev)
    y = nops.conv3x3_wino2_bias_act(x, wud, bias, cout, cpad, 0.1)
    for _ in range(3): nops.conv3x3_wino2_bias_act(x, wud, bias, cout, cpad, 0.1)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): nops.conv3x3_wino2_bias_act(x, wud, bias, cout, cpad, 0.1)
    e1.record(); torch.cuda.synchronize()
    print(f"b={b} {h}x{w} {cin}->{cout}: {e0.elapsed_time(e1) * 100:.1f} us", flush=True)
    out["in"][f"{b}x{h}x{w}x{cin}->{cout}"] = (x.cpu(), k, bias.cpu())
    out["out"][f"{b}x{h}x{w}x{cin}->{cout}"] = y.cpu()
torch.save(out, a.save)
