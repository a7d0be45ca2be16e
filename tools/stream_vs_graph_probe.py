"""How do small kernels fare beside chip-filling ones -- as plain stream launches, as hipGraph replays, with a high-priority
stream?  (Round 3: plain launches on both sides: the chain is hardly slowed; a graph replay on EITHER side: 2x slower;
stream priority and GPU_MAX_HW_QUEUES change nothing.  profiles/r03_stream_vs_graph_probe.txt)
Stream A: a queue of level-1 128->128 bf16-split Winograd layers (960 workgroups, ~130 us each); stream B (priority p): a
chain of 30 small-map convolutions (level 5: 12x40, 128->128, 60 workgroups, ~10 us each alone)."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m4depth_amd import network_ops as nops
dev = torch.device("cuda:0")
def layer(h, w, cin, cout, small):
    x = torch.randn(1, h, w, cin, device=dev); k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.zeros(cout, device=dev)
    if small:
        wp, cpad = nops.pack_conv_weights_small6(k.numpy()); wpd = torch.from_numpy(wp.view("int16")).to(dev)
        return lambda: nops.conv3x3_small6_bias_act(x, wpd, bias, cout, cpad, 0.1)
    wp, cpad = nops.pack_conv_weights_wino6(k.numpy()); wpd = torch.from_numpy(wp.view("int16")).to(dev)
    return lambda: nops.conv3x3_wino6_bias_act(x, wpd, bias, cout, cpad, 0.1)
big, small = layer(192, 640, 128, 128, False), layer(12, 40, 128, 128, True)
for _ in range(3): big(); small()
torch.cuda.synchronize()
def chain_time(prio, with_big, graph, big_graph=False):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)
    g = gb = None
    if big_graph:
        with torch.cuda.stream(sa):
            big()
        torch.cuda.synchronize()
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb, stream=sa):
            for _ in range(12): big()
    if graph:
        with torch.cuda.stream(sb):
            for _ in range(3): small()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=sb):
            for _ in range(30): small()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if with_big:
            with torch.cuda.stream(sa):
                if gb is not None: gb.replay()
                else:
                    for _ in range(12): big()
        torch.cuda._sleep(200000)
        with torch.cuda.stream(sb):
            sb.wait_stream(torch.cuda.current_stream())
            e0.record(sb)
            if g is not None: g.replay()
            else:
                for _ in range(30): small()
            e1.record(sb)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
def big_time(with_chain):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sa):
            e0.record(sa)
            for _ in range(12): big()
            e1.record(sa)
        if with_chain:
            with torch.cuda.stream(sb):
                for _ in range(90): small()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
for graph in (False, True):
    print(f"{'graph replay' if graph else 'eager launches'} of the 30-kernel chain:")
    print(f"   alone: {chain_time(0, False, graph):8.1f} us")
    for prio in (0, -1):
        print(f"   beside the Winograd queue, chain stream priority {prio:2d}: {chain_time(prio, True, graph):8.1f} us", flush=True)
print("big kernels as ONE graph replay on their stream:")
for graph in (False, True):
    print(f"   {'graph replay' if graph else 'eager launches'} of the chain beside it: {chain_time(0, True, graph, True):8.1f} us", flush=True)
print(f"12 Winograd layers (eager): alone {big_time(False):8.1f} us, with an eager 90-kernel chain beside them {big_time(True):8.1f} us")
import os
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
