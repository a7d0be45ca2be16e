import os, sys, torch
sys.path.insert(0, "/root/repo")
import m4depth_amd as M
dev = torch.device("cuda:0")
for (h, w, C, k) in [(6, 20, 192, 8), (12, 40, 128, 4), (24, 80, 96, 4), (48, 160, 64, 2), (96, 320, 32, 2), (192, 640, 16, 1)]:
    x = torch.randn(1, h, w, C, device=dev)
    for _ in range(3): y = M.cost_volume(x, x, 3, nbre_cuts=k)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = M.cost_volume(x, x, 3, nbre_cuts=k)
    e1.record(); torch.cuda.synchronize()
    print(f"{h}x{w} C={C} k={k}: {e0.elapsed_time(e1) * 50:.1f} us  checksum {y.double().sum().item():.6f}", flush=True)
