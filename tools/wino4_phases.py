"""Per-phase cycle stamps of Winograd kernel 4 (m4d_wino_set_stamps): where a chunk's time goes for wave 0."""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
from m4depth_amd._lib import lib
dev = torch.device("cuda:0")
h, w, cin, cout = 192, 640, 128, 128
x = torch.randn(1, h, w, cin, device=dev)
k = torch.randn(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
bias = torch.zeros(cout, device=dev)
wu, cpad = nops.pack_conv_weights_winograd(k.numpy(), chunk=8); wud = torch.from_numpy(wu).to(dev)
for _ in range(3): nops.conv3x3_wino2_bias_act(x, wud, bias, cout, cpad, 0.1)
buf = torch.zeros(512 * 202, dtype=torch.int64, device=dev)
lib.m4d_wino_set_stamps(ctypes.c_void_p(buf.data_ptr()))
nops.conv3x3_wino2_bias_act(x, wud, bias, cout, cpad, 0.1)
torch.cuda.synchronize()
lib.m4d_wino_set_stamps(None)
s = buf.cpu().numpy().reshape(512, 202).astype(np.int64)
n_ch = cin // 8
c = s[:, 2:2 + 5 * n_ch].reshape(512, n_ch, 5)
f = lambda a: f"median {np.median(a):8.0f}  mean {a.mean():8.0f}  p90 {np.percentile(a, 90):8.0f}"
print("workgroup total                              ", f(s[:, 201] - s[:, 0]))
print("prologue (start -> first chunk)              ", f(c[:, 0, 0] - s[:, 0]))
print("K loop                                       ", f(s[:, 1] - c[:, 0, 0]))
print("epilogue                                     ", f(s[:, 201] - s[:, 1]))
print("per chunk: A reads + raw reads + groups 0, 1  ", f(c[:, :, 1] - c[:, :, 0]))
print("per chunk: transform + V writes               ", f(c[:, :, 2] - c[:, :, 1]))
print("per chunk: group 2                            ", f(c[:, :, 3] - c[:, :, 2]))
print("per chunk: commit + loads + group 3           ", f(c[:, :, 4] - c[:, :, 3]))
print("per chunk: barrier (end -> next start)        ", f(c[:, 1:, 0] - c[:, :-1, 4]))
print("phase period                                  ", f(c[:, 1:, 0] - c[:, :-1, 0]))
print("one workgroup, chunks 4..8:\n", (c[7, 4:9, :] - c[7, 4, 0]))
