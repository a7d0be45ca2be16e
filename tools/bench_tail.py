"""The fused refiner tail (csrc/m4d_tail.hip) alone on one level's map: us per launch, for quick timing and rocprofv3 passes."""
import argparse, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--h", type=int, default=192)
ap.add_argument("--w", type=int, default=640); ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--split", action="store_true", help="the bf16-split kernel (m4d_tail6.hip) instead of the fp32-MFMA one")
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
x = torch.relu(torch.randn(a.batch, a.h, a.w, 32, device=dev)) - 0.03
k6 = (rng.standard_normal([3, 3, 32, 16]) * (2.0 / (9 * 32)) ** 0.5).astype(np.float32)
k7 = (rng.standard_normal([3, 3, 16, 5]) * (2.0 / (9 * 16)) ** 0.5).astype(np.float32)
if a.split:
    w6p, w7p = nops.pack_refiner_tail_weights6(k6, k7)
    w6d, w7d = torch.from_numpy(w6p.view(np.int16)).to(dev), torch.from_numpy(w7p.view(np.int16)).to(dev)
else:
    w6p, w7p = nops.pack_refiner_tail_weights(k6, k7)
    w6d, w7d = torch.from_numpy(w6p).to(dev), torch.from_numpy(w7p).to(dev)
b6 = torch.zeros(16, device=dev); b7 = torch.zeros(5, device=dev)
rot = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev).repeat(a.batch, 1)
trans = torch.tensor([[0.3, 0.1, 0.02]], device=dev).repeat(a.batch, 1)
cam = {"f": torch.tensor([[0.5 * a.w, 0.5 * a.h]], device=dev).repeat(a.batch, 1),
       "c": torch.tensor([[0.5 * a.w, 0.5 * a.h]], device=dev).repeat(a.batch, 1)}
state = torch.empty(a.batch, a.h, a.w, 1, device=dev)
tail = nops.refiner_tail6 if a.split else nops.refiner_tail
fn = lambda: tail(x, w6d, b6, w7d, b7, rot, trans, cam, 0.25, state)
for _ in range(3): fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()                                  # replayed from a graph: the host (3 allocations + ctypes per call) is out of the timing
with torch.cuda.graph(g):
    for _ in range(a.iters): fn()
g.replay(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
g.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
px = a.batch * a.h * a.w
name = "refiner tail (bf16 split)" if a.split else "refiner tail"
print(f"{name} {a.h}x{a.w} b={a.batch}: {us:.1f} us/launch ({us / a.batch:.1f} per frame), {px * 160 / us / 1e6:.2f} TB/s algorithmic")
