"""m4d_conv3x3_lat_chain (one launch for the refiner layers 1-5 of a coarse level) against the five separate m4d_conv3x3_lat
launches, both replayed from a hipGraph: us per level, by workers per XCD."""
import os, sys, torch
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
dev = torch.device("cuda:0")


def timed(fn, reps=10):
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


for (h, w, cin0) in [(6, 20, 472), (12, 40, 240), (24, 80, 240)]:
    chans = [cin0, 128, 128, 96, 64, 32]
    ws = [torch.from_numpy(nops.pack_conv_weights_lat((np.random.randn(3, 3, ci, co) * (2.0 / (9 * ci)) ** 0.5).astype(np.float32)).view(np.int16)).to(dev)
          for ci, co in zip(chans[:-1], chans[1:])]
    bs = [torch.zeros(co, device=dev) for co in chans[1:]]
    cfgs = [nops.lat_config(1, h, w, ci, co, final=(i == 4)) for i, (ci, co) in enumerate(zip(chans[:-1], chans[1:]))]
    x0 = torch.randn(1, h, w, cin0, device=dev)

    def separate():
        x = x0
        for i in range(5):
            x = nops.conv3x3_lat(x, ws[i], bs[i], chans[i + 1], 0.1, config=cfgs[i])
        return x
    layers = [(ws[i], bs[i], chans[i + 1], 0.1, cfgs[i]) for i in range(5)]
    row = [f"separate {timed(separate):6.1f} us"]
    for wg in (32, 16, 8, 4):
        nops.lat_chain_workgroups = wg
        row.append(f"chain/{wg} {timed(lambda: nops.conv3x3_lat_chain(x0, layers, key=('bench', h))[0]):6.1f}")
    print(f"{h}x{w}: " + "  ".join(row), flush=True)
