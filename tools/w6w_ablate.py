"""Timing ablations of the wide bf16-split Winograd kernel (build with `make W6FLAGS=-DM4D_W6W_ABLATIONS`): which part of the
K loop bounds it?  Results of the ablated kernels are wrong by construction; only the time matters."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
from m4depth_amd._lib import require_experiments
require_experiments("tools/w6w_ablate.py")
from m4depth_amd._lib import lib
raw = ctypes.CDLL(os.path.join(ROOT, "m4depth_amd", "libm4depth_hip.so"))
dev = torch.device("cuda:0")
NAMES = {0: "full kernel", 1: "no V side work (read_t + split)", 2: "no B DMA", 4: "no raw DMA", 6: "no DMA at all", 8: "no barrier",
         16: "no B fragment LDS reads", 32: "no MFMA", 17: "no V work, no B reads", 23: "no V, no DMA, no B reads (MFMA + barrier)",
         31: "MFMAs only", 33: "no V work, no MFMA", 38: "no DMA, no MFMA"}


def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


for (h, w, cin, cout) in [(192, 640, 128, 128), (48, 160, 128, 128)]:
    x = torch.randn(1, h, w, cin, device=dev)
    k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    wu6, cpad6 = nops.pack_conv_weights_wino6(k.numpy()); wud6 = torch.from_numpy(wu6.view("int16")).to(dev)
    f6 = lambda: nops.conv3x3_wino6_bias_act(x, wud6, bias, cout, cpad6, 0.1)
    lib.m4d_wino6_set_variant(1); print(f"{h}x{w} {cin}->{cout}: narrow kernel {timed(f6):7.1f} us")
    lib.m4d_wino6_set_variant(2)
    for mask, name in NAMES.items():
        raw.m4d_wino6w_set_ablation(mask)
        print(f"   ablation {mask:2d} ({name}): {timed(f6):7.1f} us", flush=True)
    raw.m4d_wino6w_set_ablation(0); lib.m4d_wino6_set_variant(0)

# ---- phase stamps (s_memtime, 100 MHz on gfx950? printed raw and as a share): per wave, first workgroups of the level-1 layer
import numpy as np
for (h, w, cin, cout) in [(48, 160, 128, 128), (192, 640, 128, 128)]:
    x = torch.randn(1, h, w, cin, device=dev)
    k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    wu6, cpad6 = nops.pack_conv_weights_wino6(k.numpy()); wud6 = torch.from_numpy(wu6.view("int16")).to(dev)
    buf = torch.zeros(64 * 8 * 16, dtype=torch.int64, device=dev)
    lib.m4d_wino6_set_variant(2)
    nops.conv3x3_wino6_bias_act(x, wud6, bias, cout, cpad6, 0.1); torch.cuda.synchronize()
    raw.m4d_wino6w_set_stamps(ctypes.c_void_p(buf.data_ptr()))
    nops.conv3x3_wino6_bias_act(x, wud6, bias, cout, cpad6, 0.1); torch.cuda.synchronize()
    raw.m4d_wino6w_set_stamps(None); lib.m4d_wino6_set_variant(0)
    st = buf.cpu().numpy().reshape(64, 8, 16).astype(np.int64)
    names = ["start", "p0 prologue DMAs issued", "p0 landed+barrier", "p0 V(0) ready", "p0 K loop done", "p0 drained+sync", "p0 epilogue done",
             "p1 prologue DMAs issued", "p1 landed+barrier", "p1 V(0) ready", "p1 K loop done", "p1 drained+sync", "p1 epilogue done"]
    print(f"{h}x{w} {cin}->{cout}: phase durations in s_memtime ticks, median over the first 30 workgroups x 8 waves (wave 0 of workgroup 0 in brackets)")
    d = np.diff(st[:30, :, :13], axis=2)
    for i, nm in enumerate(names[1:]):
        print(f"   -> {nm:28s} {np.median(d[:, :, i]):9.0f}   [{d[0, 0, i]}]")
    print(f"   total {np.median(st[:30, :, 12] - st[:30, :, 0]):9.0f} ticks")
