import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import m4depth_amd as M
from m4depth_amd import network as net, synthetic as S
dev = torch.device("cuda:0")
model = M.M4Depth(nbre_levels=6).load_numpy_weights(S.init_weights(6, seed=42), dev)
samples, cam = S.make_sequence(1, 4, 384, 1280, seed=1)
data = {k: torch.from_numpy(np.stack([s[k] for s in samples], axis=1)).to(dev) for k in ("depth", "RGB_im", "rot", "trans")}
data["new_traj"] = torch.from_numpy(np.stack([s["new_traj"] for s in samples], axis=1))
data["camera"] = {k: torch.from_numpy(v).to(dev) for k, v in cam.items()}
for _ in range(2):
    model.test_step(data)
for prio in (True, False):
    r = net.TaskGraphSequence(model, data, use_priorities=prio)
    for _ in range(3):
        r(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        r(data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"priorities={prio}: host issue {1e3 * (t1 - t0) / 10:.2f} ms/step, total {1e3 * (t2 - t0) / 10:.2f} ms/step")
    # cost of one small graph replay alone
    g = r.tasks[(1, 5)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"   level-6 task graph alone: host {1e6 * (t1 - t0) / 50:.1f} us/replay, total {1e6 * (t2 - t0) / 50:.1f} us/replay")

# ---- per-task begin / end times of one step on the real streams
r = net.TaskGraphSequence(model, data, use_priorities=False)
for _ in range(3):
    r(data)
torch.cuda.synchronize()
T, L = r.seq_len, r.n_lvls
evb = {k: torch.cuda.Event(enable_timing=True) for k in r.tasks}
eve = {k: torch.cuda.Event(enable_timing=True) for k in r.tasks}
e0 = torch.cuda.Event(enable_timing=True); e_enc = torch.cuda.Event(enable_timing=True); e_end = torch.cuda.Event(enable_timing=True)
main = r.main
with torch.cuda.stream(main):
    e0.record(main)
    r.g_enc.replay()
    e_enc.record(main)
    r.ev_enc.record(main)
for st in r.streams:
    st.wait_event(r.ev_enc)
for diag in range(T + L - 1):
    for t in range(max(0, diag - L + 1), min(T, diag + 1)):
        lvl = L - 1 - (diag - t)
        st = r.streams[t]
        with torch.cuda.stream(st):
            if t > 0:
                st.wait_event(r.ev[(t - 1, lvl)])
            evb[(t, lvl)].record(st)
            r.tasks[(t, lvl)].replay()
            eve[(t, lvl)].record(st)
            r.ev[(t, lvl)].record(st)
for t in range(T):
    main.wait_event(r.ev[(t, 0)])
with torch.cuda.stream(main):
    r.g_out.replay()
    e_end.record(main)
torch.cuda.synchronize()
print(f"encoder done at {e0.elapsed_time(e_enc) * 1e3:.0f} us, step end {e0.elapsed_time(e_end) * 1e3:.0f} us")
for t in range(T):
    print(f"frame {t}: " + "  ".join(f"L{lvl + 1} {e0.elapsed_time(evb[(t, lvl)]) * 1e3:.0f}-{e0.elapsed_time(eve[(t, lvl)]) * 1e3:.0f}" for lvl in range(L - 1, -1, -1)))
