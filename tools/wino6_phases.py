"""Cycle stamps of the bf16-split Winograd kernel (m4d_wino6_set_stamps): per workgroup start / end of the K loop / end of the
epilogue, for the level-1 and level-2 128->128 layers.  The stamped kernel is the product kernel + three s_memtime reads."""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
from m4depth_amd._lib import lib
dev = torch.device("cuda:0")
for (h, w, cin, cout) in [(192, 640, 128, 128), (96, 320, 128, 128), (192, 640, 64, 128)]:
    x = torch.randn(1, h, w, cin, device=dev)
    k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.zeros(cout, device=dev)
    wu6, cpad = nops.pack_conv_weights_wino6(k.numpy()); wud = torch.from_numpy(wu6.view("int16")).to(dev)
    for _ in range(3): nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1)
    st = torch.zeros(64 * 1280, dtype=torch.int64, device=dev)
    lib.m4d_wino6_set_stamps(ctypes.c_void_p(st.data_ptr()))
    nops.conv3x3_wino6_bias_act(x, wud, bias, cout, cpad, 0.1)
    torch.cuda.synchronize()
    lib.m4d_wino6_set_stamps(None)
    full = st.view(64, 1280).cpu().double()
    nwg = min(64, -(-h // 16) * -(-w // 16) * (cpad // 64))
    full = full[:nwg]
    hdr = full[:, 1024:1027]
    kl, ep = hdr[:, 1] - hdr[:, 0], hdr[:, 2] - hdr[:, 1]
    npos = min(32, cin // 4)
    pos = full[:, :1024].view(nwg, 8, 32, 4)[:, :, :npos]                       # [wg][wave][position][after barrier, -, before wait, after wait]
    work, wait = pos[..., 2] - pos[..., 0], pos[..., 3] - pos[..., 2]
    bar = pos[:, :, 1:, 0] - pos[:, :, :-1, 3]
    per = pos[:, :, 1:, 0] - pos[:, :, :-1, 0]
    print(f"{h}x{w} {cin}->{cout}: prologue + K loop {kl.mean():.0f} ticks ({kl.mean() / (cin // 16):.0f} per 16-channel chunk), epilogue {ep.mean():.0f}; "
          f"position period {per.mean():.0f}")
    for wvi in range(8):
        print(f"    wave {wvi} (row {wvi & 3}, M-tile {wvi >> 2}): MFMAs + A generation + DMA issue {work[:, wvi].mean():.0f}, "
              f"vmcnt/lgkmcnt wait {wait[:, wvi].mean():.0f}, barrier {bar[:, wvi].mean():.0f}; by column "
              + ", ".join(f"{work[:, wvi, c::4].mean():.0f}" for c in range(4)))
