"""The row-split variants of the r = 3 SNCV kernel (M4D_SNCV_YS = 1 | 2 | 4, read once per process) and the small-map
kernel (M4D_SNCV_SMALL_PX) must give the same bits: run once per setting with --save, then --compare.
The inputs are saved next to the outputs, so the test can also compare every variant with the CPU oracle directly.
Used by tests/test_gpu_ops.py::test_sncv_variants_are_bitwise_identical."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--save"); ap.add_argument("--compare", nargs="+")
a = ap.parse_args()
if a.compare:
    ref = torch.load(a.compare[0])["out"]
    for other in a.compare[1:]:
        y = torch.load(other)["out"]
        for k in ref:
            same = torch.equal(ref[k].view(torch.int32), y[k].view(torch.int32))
            print(os.path.basename(other), k, "bit-identical" if same else f"DIFFERENT max {(ref[k] - y[k]).abs().max().item():.3e}")
    sys.exit(0)
import m4depth_amd as M
from m4depth_amd import network_ops as nops
dev = torch.device("cuda:0")
out = {"in": {}, "out": {}}
g = torch.Generator().manual_seed(5)
for (b, h, w, C, k) in [(1, 48, 160, 64, 2), (2, 37, 53, 32, 2), (1, 96, 100, 16, 1), (1, 24, 80, 96, 4), (3, 9, 11, 192, 8)]:
    x = nops.normalize_cuts(torch.randn(b, h, w, C, generator=g).to(dev), k)
    out["in"][f"{b}x{h}x{w}x{C}/{k}"] = x.cpu()
    out["out"][f"{b}x{h}x{w}x{C}/{k}"] = M.cost_volume(x, x, 3, nbre_cuts=k).cpu()
torch.save(out, a.save)
