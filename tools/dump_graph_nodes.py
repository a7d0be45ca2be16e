"""The kernel nodes of the captured sequence forward (network.GraphedSequence, what bench.py replays) at a batch size, from the
HIP runtime's own dump of the instantiated graph (DEBUG_HIP_GRAPH_DOT_PRINT=1: graph_* files in the working directory; the
largest is the sequence graph): DOT file + one line per distinct kernel; exits non-zero if any node is not one of
libm4depth_hip.so's kernels.  python tools/dump_graph_nodes.py --batch 32 --out profiles/x.dot"""
import argparse, collections, glob, os, re, shutil, sys, tempfile
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--height", type=int, default=384)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--out", required=True)
a = ap.parse_args()
out = os.path.abspath(a.out)
os.environ["DEBUG_HIP_GRAPH_DOT_PRINT"] = "1"            # read by the HIP runtime when it instantiates a graph
os.environ.setdefault("M4D_STAGGER_AUTOTUNE", "0")       # one capture (the two forms differ in a launch argument, not in nodes)
work = tempfile.mkdtemp(prefix="m4d_dot_")
os.chdir(work)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import m4depth_amd as M
from m4depth_amd import network as net, synthetic as S

dev = torch.device("cuda:0")
model = M.M4Depth(nbre_levels=a.levels)
model.load_numpy_weights(S.init_weights(a.levels, seed=42), dev)
samples, cam = S.make_sequence(a.batch, a.frames, a.height, a.width, seed=1)
d = {k: torch.stack([torch.from_numpy(s[k]).to(dev) for s in samples], dim=1) for k in ("depth", "RGB_im", "rot", "trans")}
d["new_traj"] = torch.stack([torch.from_numpy(s["new_traj"]) for s in samples], dim=1)
d["camera"] = {k: torch.from_numpy(v).to(dev) for k, v in cam.items()}
model.compile(metrics=M.default_metrics())
model.test_step(d)
runner = net.GraphedSequence(model, d)
runner(d)
torch.cuda.synchronize()
best, names = None, []
for f in glob.glob(os.path.join(work, "*")):
    try:
        n = re.findall(r"_Z\w+", open(f).read())
    except Exception:
        continue
    if len(n) > len(names):
        best, names = f, n
if best is None:
    sys.exit("dump_graph_nodes: the HIP runtime wrote no graph dump (DEBUG_HIP_GRAPH_DOT_PRINT)")
shutil.copy(best, out)
cnt = collections.Counter(names)
foreign = [n for n in cnt if not (n.startswith("_ZN12_GLOBAL__N_1") or "conv3x3_wino" in n)]
print(f"batch {a.batch}, {a.height}x{a.width}, {a.levels} levels, {a.frames} frames: {len(names)} kernel nodes, {len(cnt)} distinct kernels, "
      f"{len(foreign)} not from libm4depth_hip.so; {int(net.lib.m4d_launch_count())} library launches since load")
for n, c in cnt.most_common():
    print(f"  {c:4d}  {n[:120]}")
shutil.rmtree(work, ignore_errors=True)
sys.exit(1 if foreign else 0)
