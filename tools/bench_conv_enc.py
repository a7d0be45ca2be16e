"""Encoder layers: MFMA conv (stride 1/2) vs MIOpen (+epilogue/pad), batch = frames of a sequence."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
from tools.bench_conv import timeit
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
h, w, cin = 384, 1280, 3
for co in [16, 32, 64, 96, 128, 192]:
    for stride in (1, 2):
        ci = cin if stride == 1 else co
        x = torch.randn(b, h, w, ci, device=dev)
        k = torch.randn(3, 3, ci, co) * (2.0 / (9 * ci)) ** 0.5
        bias = torch.randn(co, device=dev) * 0.1
        wp, cpad = nops.pack_conv_weights(k.numpy()); wpd = torch.from_numpy(wp).to(dev)
        t_m = timeit(lambda: nops.conv3x3_bias_act(x, wpd, bias, co, cpad, 0.1, stride=stride))
        wt = k.permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last).to(dev)
        def mi():
            xx = x.permute(0, 3, 1, 2)
            if stride == 2: xx = F.pad(xx, (0, 1, 0, 1))
            y = F.conv2d(xx, wt, None, stride, 1 if stride == 1 else 0).permute(0, 2, 3, 1)
            return nops.bias_act_(y if y.is_contiguous() else y.contiguous(), bias, 0.1)
        t_i = timeit(mi)
        fl = 2 * 9 * ci * co * (h // stride) * (w // stride) * b
        print(f"{h}x{w} b={b} {ci:3d}->{co:3d} s{stride}: mfma {t_m:8.1f} us ({fl/t_m/1e6:5.1f} TF/s) | miopen {t_i:8.1f} us | {t_i/t_m:.2f}x", flush=True)
    h //= 2; w //= 2; cin = co
