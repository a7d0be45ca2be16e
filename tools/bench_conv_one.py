"""One layer of the hand-written MFMA convolution (level-1 refiner 128->128 by default), many
identical launches: for rocprofv3 --pmc passes and quick timing."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--cin", type=int, default=128)
ap.add_argument("--cout", type=int, default=128); ap.add_argument("--h", type=int, default=192)
ap.add_argument("--w", type=int, default=640); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--winograd", type=int, default=0, help="0 = direct MFMA kernel, 1 / 2 = fp32-MFMA Winograd kernel 1 / 2(4), 6 = the bf16-split Winograd kernel")
ap.add_argument("--variant", type=int, default=0, help="(experiments build) m4d_wino6_set_variant: 2 = the wide kernel, 3 = the half-tile kernel")
ap.add_argument("--kernel", type=int, default=0, help="m4d_conv3x3_wino6_bias_act_k: 0 = the library's choice, 1 = one workgroup per unit, 2 = persistent")
a = ap.parse_args()
if a.variant:
    from m4depth_amd._lib import lib
    lib.m4d_wino6_set_variant(a.variant)
dev = torch.device("cuda:0")
x = torch.randn(a.batch, a.h, a.w, a.cin, device=dev)
k = torch.randn(3, 3, a.cin, a.cout) * (2.0 / (9 * a.cin)) ** 0.5
bias = torch.randn(a.cout, device=dev) * 0.1
if a.winograd == 6:
    wp, cpad = nops.pack_conv_weights_wino6(k.numpy()); wp = wp.view("int16")
    fn = lambda *args_: nops.conv3x3_wino6_bias_act(*args_, kernel=a.kernel)
elif a.winograd:
    wp, cpad = nops.pack_conv_weights_winograd(k.numpy(), chunk=8 if a.winograd == 2 else 16)
    fn = nops.conv3x3_wino2_bias_act if a.winograd == 2 else nops.conv3x3_wino_bias_act
else:
    wp, cpad = nops.pack_conv_weights(k.numpy())
    fn = nops.conv3x3_bias_act
wpd = torch.from_numpy(wp).to(dev)
for _ in range(3): fn(x, wpd, bias, a.cout, cpad, 0.1)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters): fn(x, wpd, bias, a.cout, cpad, 0.1)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
fl = 2 * 9 * a.cin * a.cout * a.h * a.w * a.batch
print(f"conv {a.cin}->{a.cout} {a.h}x{a.w} b={a.batch}: {us:.1f} us/launch, {fl / us / 1e6:.1f} TFLOP/s")
