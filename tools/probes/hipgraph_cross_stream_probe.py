"""Minimal hipGraph probe: chain X of n small kernels on one stream; chain Y on another stream.
variant 'all': Y_i waits for X_{8+i};  'one': only Y_0 waits (for X_8);  'tail': Y_0 waits for X_8, X has NO other events."""
import sys, torch, time
cyc = 20000
variant = sys.argv[1]; n = int(sys.argv[2]); mode = sys.argv[3]
dev = torch.device("cuda:0")
main = torch.cuda.Stream(); sx = torch.cuda.Stream(); sy = torch.cuda.Stream()
def body():
    fork = torch.cuda.Event(); fork.record(main)
    sx.wait_event(fork)
    evs = {}
    with torch.cuda.stream(sx):
        for i in range(n):
            torch.cuda._sleep(cyc)
            if variant == "all" or i == 8:
                e = torch.cuda.Event(); e.record(sx); evs[i] = e
    with torch.cuda.stream(sy):
        for i in range(32):
            if (variant == "all" and 8 + i in evs) or i == 0:
                sy.wait_event(evs[8 + i] if variant == "all" else evs[8])
            torch.cuda._sleep(cyc // 2)
        ey = torch.cuda.Event(); ey.record(sy)
    ex = torch.cuda.Event(); ex.record(sx)
    main.wait_event(ex); main.wait_event(ey)
    return evs, fork, ex, ey
with torch.cuda.stream(main):
    keep = body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=main):
    keep2 = body()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
for _ in range(6):
    g.replay()
    if mode == "sync": torch.cuda.synchronize()
torch.cuda.synchronize()
