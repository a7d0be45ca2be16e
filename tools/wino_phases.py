"""Per-phase cycle breakdown of the Winograd convolution (m4d_wino_set_stamps): where a workgroup's time goes."""
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
V2 = len(sys.argv) > 1 and sys.argv[1] == "2"
chunk = 8 if V2 else 16
conv = nops.conv3x3_wino2_bias_act if V2 else nops.conv3x3_wino_bias_act
wu, cpad = nops.pack_conv_weights_winograd(k.numpy(), chunk=chunk); wud = torch.from_numpy(wu).to(dev)
for _ in range(3): conv(x, wud, bias, cout, cpad, 0.1)
buf = torch.zeros(512 * 202, dtype=torch.int64, device=dev)
lib.m4d_wino_set_stamps(ctypes.c_void_p(buf.data_ptr()))
conv(x, wud, bias, cout, cpad, 0.1)
torch.cuda.synchronize()
lib.m4d_wino_set_stamps(None)
s = buf.cpu().numpy().reshape(512, 202).astype(np.int64)
n_ch = cin // chunk
tot = s[:, 1] - s[:, 0]
c = s[:, 2:2 + 5 * n_ch].reshape(512, n_ch, 5)
commit = c[:, :, 1] - c[:, :, 0]          # commit + raw-load issue + first barrier
trans = c[:, :, 2] - c[:, :, 1]
bar2 = c[:, :, 3] - c[:, :, 2]
mfma = c[:, :, 4] - c[:, :, 3]
gap = c[:, 1:, 0] - c[:, :-1, 4]
pro = c[:, 0, 0] - s[:, 0]
epi = s[:, 1] - c[:, -1, 4]
f = lambda a: f"median {np.median(a):8.0f}  mean {a.mean():8.0f}  p90 {np.percentile(a, 90):8.0f}"
print("cycles (100 MHz counter x clock ratio? see total): workgroup total      ", f(tot))
print("  prologue (first loads)                     ", f(pro))
print("  per chunk: commit + load issue + barrier 1  ", f(commit))
print("  per chunk: input transform                  ", f(trans))
print("  per chunk: barrier 2                        ", f(bar2))
print("  per chunk: 4 positions x NT x 8 MFMA        ", f(mfma))
print("  epilogue                                    ", f(epi))
print("  sum of chunk phases / total                 ", ((commit + trans + bar2 + mfma).sum(axis=1) / tot).mean())
