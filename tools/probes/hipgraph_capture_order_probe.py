"""hipGraph probe 2: does the CAPTURE ORDER of a cross-stream child change when it may start?
'late': X_0..X_39 captured, then Y (waits for X_8);  'early': X_0..X_8, then Y's wait + all of Y, then X_9..X_39;
'split': X_0..X_8 on stream sx, X_9.. on a THIRD stream (waits ev8), Y on sy (waits ev8)."""
import sys, torch
cyc = 20000
variant = sys.argv[1]; n = 40
main = torch.cuda.Stream(); sx = torch.cuda.Stream(); sy = torch.cuda.Stream(); sz = torch.cuda.Stream()
def body():
    keep = []
    fork = torch.cuda.Event(); fork.record(main); keep.append(fork)
    sx.wait_event(fork)
    def xk(i0, i1, st):
        with torch.cuda.stream(st):
            for i in range(i0, i1): torch.cuda._sleep(cyc)
    def yk():
        with torch.cuda.stream(sy):
            sy.wait_event(ev8)
            for i in range(32): torch.cuda._sleep(cyc // 2)
            ey = torch.cuda.Event(); ey.record(sy); keep.append(ey)
            return ey
    xk(0, 9, sx)
    ev8 = torch.cuda.Event(); ev8.record(sx); keep.append(ev8)
    if variant == "late":
        xk(9, n, sx); ey = yk(); last = sx
    elif variant == "early":
        ey = yk(); xk(9, n, sx); last = sx
    else:
        sz.wait_event(ev8); ey = yk(); xk(9, n, sz); last = sz
    ex = torch.cuda.Event(); ex.record(last); keep.append(ex)
    main.wait_event(ex); main.wait_event(ey)
    if last is not sx:
        e2 = torch.cuda.Event(); e2.record(sx); keep.append(e2); main.wait_event(e2)
    return keep
with torch.cuda.stream(main):
    k1 = body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=main):
    k2 = body()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
for _ in range(6):
    g.replay(); torch.cuda.synchronize()
