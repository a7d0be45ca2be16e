"""Host cost of the segmented step's launches: per-segment hipGraphLaunch time on the host, steady-state step period on the GPU."""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
import m4depth_amd as M
from m4depth_amd import network as net, synthetic as S

dev = torch.device("cuda:0")
args = types.SimpleNamespace(gpus=1, batch=1, seq_len=4, height=384, width=1280, levels=6, dscv_range=4, sncv_range=3)
weights = S.init_weights(6, seed=42)
model = M.M4Depth(nbre_levels=6)
model.load_numpy_weights(weights, dev)
model.compile(metrics=M.default_metrics())
data = B.make_batch(args, 0, dev, torch)
model.test_step(data)
for kind in ("whole", "segmented"):
    cls = net.GraphedSequence if kind == "whole" else net.SegmentedSequence
    runner = cls(model, data, warmup=1, autotune=False)
    data.update(runner.input_buffers())
    host = {}
    if kind == "segmented":
        def timed_issue(n):
            t = time.perf_counter(); runner.graph[n].replay(); host.setdefault(n, []).append(time.perf_counter() - t)
        replay = lambda: runner._play(timed_issue)
    else:
        def replay():
            t = time.perf_counter(); runner.graph.replay(); host.setdefault("g", []).append(time.perf_counter() - t)
    for _ in range(30):
        replay()
    torch.cuda.synchronize()
    host.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(100):
        replay()
    t_issue = time.perf_counter() - t0
    e1.record(); e1.synchronize()
    print(f"{kind}: GPU {e0.elapsed_time(e1) / 100 * 1e3:.1f} us/step, host issue {t_issue / 100 * 1e6:.1f} us/step; per launch (us, median): "
          + ", ".join(f"{n} {sorted(v)[len(v) // 2] * 1e6:.0f}" for n, v in host.items()))

# segment boundaries of the segmented step on the GPU's clock (events between the graph launches), steady state
runner = net.SegmentedSequence(model, data, warmup=1, autotune=False)
data.update(runner.input_buffers())
cur = torch.cuda.current_stream()
for _ in range(30):
    runner._replay()
torch.cuda.synchronize()
acc = {}
N = 50
marks_all = []
for _ in range(N):
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("s", "a", "b0", "b", "c", "d0", "d")}
    ev["s"].record(cur)
    runner.graph["a"].replay(); ev["a"].record(cur)
    runner._ev_a.record(cur)
    with torch.cuda.stream(runner.stream_b):
        runner.stream_b.wait_event(runner._ev_a)
        ev["b0"].record(runner.stream_b)
        runner.graph["b"].replay(); ev["b"].record(runner.stream_b)
        runner._ev_b.record(runner.stream_b)
    runner.graph["c"].replay(); ev["c"].record(cur)
    cur.wait_event(runner._ev_b)
    ev["d0"].record(cur)
    runner.graph["d"].replay(); ev["d"].record(cur)
    marks_all.append(ev)
torch.cuda.synchronize()
for k in ("a", "b0", "b", "c", "d0", "d"):
    v = sorted(e["s"].elapsed_time(e[k]) * 1e3 for e in marks_all[5:])
    print(f"  {k:3s} at {v[len(v) // 2]:8.1f} us after the step's start (median of {len(v)})")
v = sorted(marks_all[i]["s"].elapsed_time(marks_all[i + 1]["s"]) * 1e3 for i in range(5, N - 1))
print(f"  step period {v[len(v) // 2]:8.1f} us")

# which side stream runs BESIDE the calling stream?  (HIP streams share a few hardware queues; two streams on one queue serialise)
def period(n=60):
    for _ in range(20):
        runner._replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        runner._replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cands = [runner.stream_b] + [torch.cuda.Stream() for _ in range(9)]
for i, s in enumerate(cands):
    runner.stream_b = s
    print(f"  side stream candidate {i}: {period():8.1f} us/step (calling stream = default)")
own = torch.cuda.Stream()
with torch.cuda.stream(own):
    for i, s in enumerate(cands[:6]):
        runner.stream_b = s
        print(f"  side stream candidate {i}: {period():8.1f} us/step (calling stream = a stream of its own)")

runner.stream_b = torch.cuda.current_stream()
print(f"  all four segments on ONE stream (a, b, c, d in turn): {period():8.1f} us/step")
runner.stream_b = cands[1]
print(f"  side stream again: {period():8.1f} us/step")
whole = net.GraphedSequence(model, data, warmup=1, autotune=False)
def period_whole(n=60):
    for _ in range(20):
        whole.graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        whole.graph.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"  one graph: {period_whole():8.1f} us/step")

# stream priorities: the encoder batch on a LOW-priority side stream / the chain segments on a HIGH-priority calling stream
for prio in (1, -1):
    try:
        runner.stream_b = torch.cuda.Stream(priority=prio)
        print(f"  side stream priority {prio}: {period():8.1f} us/step")
    except Exception as e:
        print(f"  side stream priority {prio}: refused ({e})")
runner.stream_b = cands[1]
hi = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(hi):
    print(f"  calling stream priority -1, side stream normal: {period():8.1f} us/step")
print("least / greatest priority:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
