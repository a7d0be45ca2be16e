"""The refiner convolutions of the coarse levels on m4d_conv3x3_lat (csrc/m4d_convlat.hip): us per layer as a dependent chain
replayed from a hipGraph (no host launch overhead), the default configuration (network_ops.lat_config) beside every other
(mt, kw, s_out) and beside conv3x3_small6 -- the table the defaults were read from.  `python tools/bench_lat_convs.py [sweep]`"""
import os, sys, itertools, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
dev = torch.device("cuda:0")
SWEEP = len(sys.argv) > 1 and sys.argv[1] == "sweep"


def timed(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record(s)
            for _ in range(5):
                g.replay()
            e1.record(s); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (5 * reps))
    return best


for (h, w, cin0) in [(6, 20, 472), (12, 40, 240), (24, 80, 240), (48, 160, 128)]:
    chans = [cin0, 128, 128, 96, 64, 32]
    layers = []
    for ci, co in zip(chans[:-1], chans[1:]):
        k = (torch.randn(3, 3, ci, co) * (2.0 / (9 * ci)) ** 0.5).numpy()
        wl = torch.from_numpy(nops.pack_conv_weights_lat(k).view("int16")).to(dev)
        w6, cpad = nops.pack_conv_weights_small6(k)
        layers.append((wl, torch.from_numpy(w6.view("int16")).to(dev), cpad, torch.zeros(co, device=dev), ci, co))
    x0 = torch.randn(1, h, w, cin0, device=dev)

    def chain_lat(n, override=None):
        x = x0
        for i, (wl, _, _, bz, ci, co) in enumerate(layers[:n]):
            cfg = override if (override is not None and i == n - 1) else None
            x = nops.conv3x3_lat(x, wl, bz, co, 0.1, final=(i == 4), config=cfg)
        return x

    def chain_small6(n):
        x = x0
        for (_, w6, cpad, bz, ci, co) in layers[:n]:
            x = nops.conv3x3_small6_bias_act(x, w6, bz, co, cpad, 0.1)
        return x
    if h * w <= 2048:
        prev, row = 0.0, []
        for n in range(1, 6):
            t = timed(lambda: chain_small6(n)); row.append(t - prev); prev = t
        print(f"{h}x{w} small6 : " + ", ".join(f"{l[4]}->{l[5]} {t:5.1f}" for l, t in zip(layers, row)) + f"; chain {prev:.1f} us")
    prev, row = 0.0, []
    for n in range(1, 6):
        t = timed(lambda: chain_lat(n)); row.append(t - prev); prev = t
    cfgs = [nops.lat_config(1, h, w, l[4], l[5], final=(i == 4)) for i, l in enumerate(layers)]
    print(f"{h}x{w} lat    : " + ", ".join(f"{l[4]}->{l[5]} {t:5.1f} {c}" for l, t, c in zip(layers, row, cfgs)) + f"; chain {prev:.1f} us")
    if SWEEP:
        for n in range(1, 6):
            base = timed(lambda: chain_lat(n - 1)) if n > 1 else 0.0
            ci, co = layers[n - 1][4], layers[n - 1][5]
            nch = -(-ci // 16)
            res = []
            for mt, kw, so in itertools.product((1, 2, 4), (1, 2, 4), (1, 2, 3, 4)):
                if so > nch or (so - 1) * (-(-nch // so)) >= nch or (n == 5 and so > 1):
                    continue
                if 2 * kw * {1: 60, 2: 100, 4: 180}[mt] * 96 > 160 * 1024:
                    continue
                try:
                    res.append((timed(lambda: chain_lat(n, (mt, kw, so)), reps=10) - base, (mt, kw, so)))
                except RuntimeError as e:
                    print("   ", (mt, kw, so), "failed:", str(e)[:80])
            res.sort()
            print(f"   layer {n} {ci}->{co}: " + "  ".join(f"{c}:{t:.1f}" for t, c in res[:8]) + f"  ... worst {res[-1][1]}:{res[-1][0]:.1f}")
