"""The refiner convolutions of the three coarsest levels (one-launch small-map kernel) as a dependent chain replayed from a
hipGraph: us per layer without host launch overhead."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
dev = torch.device("cuda:0")
SPLIT = len(sys.argv) > 1 and sys.argv[1] == "split"
print("bf16-split kernel" if SPLIT else "fp32-MFMA kernel")
for (h, w, cin0) in [(6, 20, 472), (12, 40, 240), (24, 80, 240)]:
    chans = [cin0, 128, 128, 96, 64, 32]
    layers = []
    for ci, co in zip(chans[:-1], chans[1:]):
        k = torch.randn(3, 3, ci, co) * (2.0 / (9 * ci)) ** 0.5
        if SPLIT:
            wp, cpad = nops.pack_conv_weights_small6(k.numpy()); wp = wp.view("int16")
        else:
            wp, cpad = nops.pack_conv_weights(k.numpy())
        layers.append((torch.from_numpy(wp).to(dev), torch.zeros(co, device=dev), co, cpad))
    x0 = torch.randn(1, h, w, cin0, device=dev)

    def chain(n_layers):
        x = x0
        for (wp, b, co, cpad) in layers[:n_layers]:
            x = (nops.conv3x3_small6_bias_act if SPLIT else nops.conv3x3_small_bias_act)(x, wp, b, co, cpad, 0.1)
        return x
    prev = 0.0
    times = []
    for n in range(1, 6):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): chain(n)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(10): chain(n)
            g.replay(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(5): g.replay()
            e1.record(s); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / 50
        times.append(t - prev); prev = t
    print(f"{h}x{w}: " + ", ".join(f"{ci}->{co} {t:5.1f} us" for (ci, co), t in zip(zip(chans[:-1], chans[1:]), times)) + f"; chain {prev:.1f} us")
