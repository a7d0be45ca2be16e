"""Isolates what makes hipStreamEndCapture crash on a multi-stream capture (level pipeline)."""
import sys
import torch

stage = int(sys.argv[1])
dev = torch.device("cuda:0")
x = torch.ones(1 << 20, device=dev)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()
keep = []


def body():
    main = torch.cuda.current_stream()
    fork = torch.cuda.Event(); fork.record(main); keep.append(fork)
    s0.wait_event(fork); s1.wait_event(fork)
    if stage >= 5:
        s2.wait_event(fork); s3.wait_event(fork)
    outs = []
    if stage == 1:                       # independent branches
        with torch.cuda.stream(s0):
            outs.append(x * 2)
        with torch.cuda.stream(s1):
            outs.append(x * 3)
    elif stage == 2:                     # cross dependencies s0 -> s1 -> s0
        with torch.cuda.stream(s0):
            a = x * 2
            e = torch.cuda.Event(); e.record(s0); keep.append(e)
        with torch.cuda.stream(s1):
            s1.wait_event(e)
            b = a + 1
            e2 = torch.cuda.Event(); e2.record(s1); keep.append(e2)
        with torch.cuda.stream(s0):
            s0.wait_event(e2)
            outs.append(b * 2)
    elif stage == 3:                     # many events, several never waited on
        prev = None
        for i in range(12):
            st = (s0, s1)[i % 2]
            with torch.cuda.stream(st):
                if prev is not None:
                    st.wait_event(prev)
                y = x * float(i + 1)
                for _ in range(3):
                    e = torch.cuda.Event(); e.record(st); keep.append(e)
                prev = e
                outs.append(y)
    elif stage == 4:                     # wait on the same event twice / redundant waits
        with torch.cuda.stream(s0):
            a = x * 2
            e = torch.cuda.Event(); e.record(s0); keep.append(e)
        with torch.cuda.stream(s1):
            s1.wait_event(e); s1.wait_event(e); s1.wait_event(fork)
            outs.append(a + 1)
    elif stage == 5:                     # one stream per frame, dependencies flow one way: s_t -> s_{t+1}, 6 events each
        evs = {}
        for t, st in enumerate((s0, s1, s2, s3)):
            with torch.cuda.stream(st):
                y = x
                for l in range(6):
                    if t > 0:
                        st.wait_event(evs[(t - 1, l)])
                    y = y * 1.5 + float(l)
                    e = torch.cuda.Event(); e.record(st); keep.append(e)
                    evs[(t, l)] = e
                outs.append(y)
    for st in ((s0, s1) if stage < 5 else (s0, s1, s2, s3)):
        e = torch.cuda.Event(); e.record(st); keep.append(e)
        main.wait_event(e)
    return sum(o.sum() for o in outs)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = body()
g.replay()
torch.cuda.synchronize()
print("stage", stage, "ok", float(r))
