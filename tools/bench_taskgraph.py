"""GraphedSequence (one hipGraph per step, frames on captured side streams) vs TaskGraphSequence (one hipGraph per
(frame, level) task replayed on real streams, tools/taskgraph_sequence.py) at the bench configuration."""
import os, sys, time, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import m4depth_amd as M
from m4depth_amd import network as net, synthetic as S
import bench
from taskgraph_sequence import TaskGraphSequence
args = types.SimpleNamespace(batch=int(os.environ.get("B", "1")), seq_len=4, height=384, width=1280)
dev = torch.device("cuda:0")
W = S.init_weights(6, seed=42)
def timeit(step, n=20):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for name, cls, kw in (("one graph", net.GraphedSequence, {}), ("task graphs", TaskGraphSequence, {}), ("task graphs + priorities", TaskGraphSequence, {"use_priorities": True})):
    model = M.M4Depth(nbre_levels=6); model.load_numpy_weights(W, dev); model.compile(metrics=M.default_metrics())
    data = bench.make_batch(args, 0, dev, torch)
    model.test_step(data)
    runner = cls(model, data, **kw)
    dt = timeit(lambda: runner(data))
    print(f"{name:26s} {dt * 1e3:7.3f} ms/step  {args.batch * 4 / dt:8.1f} frames/s", flush=True)
