"""The kernel nodes of the captured sequence forward (network.GraphedSequence, what bench.py replays) at a batch size: DOT dump
of the hipGraph (hipGraphDebugDotPrint via torch.cuda.CUDAGraph.debug_dump) + a summary line per distinct kernel; exits
non-zero if any node is not one of libm4depth_hip.so's kernels.  python tools/dump_graph_nodes.py --batch 32 --out profiles/x.dot"""
import argparse, collections, os, re, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import m4depth_amd as M
from m4depth_amd import network as net, synthetic as S

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--out", required=True)
a = ap.parse_args()
dev = torch.device("cuda:0")
L, H, W, T = 6, 384, 1280, 4
weights = S.init_weights(L, seed=42)
model = M.M4Depth(nbre_levels=L)
model.load_numpy_weights(weights, dev)
samples, cam = S.make_sequence(a.batch, T, H, W, seed=1)
d = {k: torch.stack([torch.from_numpy(s[k]).to(dev) for s in samples], dim=1) for k in ("depth", "RGB_im", "rot", "trans")}
d["new_traj"] = torch.stack([torch.from_numpy(s["new_traj"]) for s in samples], dim=1)
d["camera"] = {k: torch.from_numpy(v).to(dev) for k, v in cam.items()}
model.compile(metrics=M.default_metrics())
model.test_step(d)
runner = net.GraphedSequence(model, d, debug_dot=a.out)
runner(d)
torch.cuda.synchronize()
names = re.findall(r"_Z\w+", open(a.out).read())
cnt = collections.Counter(names)
foreign = [n for n in cnt if not (n.startswith("_ZN12_GLOBAL__N_1") or "conv3x3_wino" in n)]
print(f"batch {a.batch}: {len(names)} kernel nodes, {len(cnt)} distinct kernels, {len(foreign)} not from libm4depth_hip.so")
for n, c in cnt.most_common():
    print(f"  {c:4d}  {n[:120]}")
sys.exit(1 if (foreign or not names) else 0)
