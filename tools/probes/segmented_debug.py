"""Debug: SegmentedSequence eager pass / replay against test_step, per (frame, level, key)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import m4depth_amd as M
from m4depth_amd import network as net, synthetic as S

dev = torch.device("cuda:0")
L, H, Wd, T, b = 3, 64, 96, 4, 1
W = S.init_weights(L, seed=9)
model = M.M4Depth(nbre_levels=L)
model.load_numpy_weights(W, dev)
model.compile(metrics=M.default_metrics())


def to_dev(x):
    if isinstance(x, dict):
        return {k: to_dev(v) for k, v in x.items()}
    return torch.from_numpy(x).to(dev)


def batch(seed):
    samples, cam = S.make_sequence(b, T, H, Wd, seed=seed)
    d = {k: torch.stack([to_dev(s[k]) for s in samples], dim=1) for k in ("depth", "RGB_im", "rot", "trans")}
    d["new_traj"] = torch.stack([torch.from_numpy(s["new_traj"]) for s in samples], dim=1)
    d["camera"] = to_dev(cam)
    return d


def snap():
    return {(f, l, k): est[k].cpu().numpy().copy() for f, frame in enumerate(model.last_estimates)
            for l, est in enumerate(frame) for k in ("depth", "parallax")}


def diff(tag, got, want):
    bad = [(key, int((got[key].view(np.uint32) != want[key].view(np.uint32)).sum()), got[key].size) for key in want
           if not np.array_equal(got[key].view(np.uint32), want[key].view(np.uint32))]
    print(tag, "OK" if not bad else bad)


d1, d2 = batch(41), batch(42)
model.test_step(d1); r1 = snap()
model.test_step(d2); r2 = snap()
model.test_step(d1); diff("eager again d1", snap(), r1)
runner = net.SegmentedSequence(model, d1, autotune=False)
torch.cuda.synchronize()
cap_est = model.last_estimates
with torch.cuda.stream(runner.stream):
    runner._run()
torch.cuda.synchronize()
diff("eager segmented pass d1 (static buffers hold d1)", snap(), r1)
model.last_estimates = cap_est
for name, d, r in (("d1", d1, r1), ("d2", d2, r2), ("d1", d1, r1)):
    model.graphed_test_step(d, runner)
    torch.cuda.synchronize()
    diff("replay " + name, snap(), r)
whole = net.GraphedSequence(model, d1, autotune=False)
for name, d, r in (("d1", d1, r1), ("d2", d2, r2)):
    model.graphed_test_step(d, whole)
    torch.cuda.synchronize()
    diff("one-graph replay " + name, snap(), r)
