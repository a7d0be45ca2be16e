"""MFMA conv (csrc/m4d_conv.hip) vs MIOpen (+ HIP bias/lrelu epilogue) per refiner layer."""
import argparse, os, sys, time
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd import network_ops as nops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--levels", default="1,2,3")
    ap.add_argument("--no-miopen", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    fin = {1: 64, 2: 122, 3: 122, 4: 238, 5: 238, 6: 470}
    chans = [128, 128, 96, 64, 32, 16, 5]
    for lvl in [int(v) for v in args.levels.split(",")]:
        h, w = 384 >> lvl, 1280 >> lvl
        cin = fin[lvl]
        tot_m, tot_o, tot_f = 0.0, 0.0, 0.0
        for co in chans:
            x = torch.randn(args.batch, h, w, cin, device=dev)
            k = (torch.randn(3, 3, cin, co) * (2.0 / (9 * cin)) ** 0.5)
            bias = torch.randn(co, device=dev) * 0.1
            wp, cpad = nops.pack_conv_weights(k.numpy())
            wpd = torch.from_numpy(wp).to(dev)
            t_mine = timeit(lambda: nops.conv3x3_bias_act(x, wpd, bias, co, cpad, 0.1))
            fl = 2 * 9 * cin * co * h * w * args.batch
            line = f"L{lvl} b={args.batch} {cin:4d}->{co:4d}: mfma {t_mine:8.1f} us {fl / t_mine / 1e6:6.1f} TF/s"
            if not args.no_miopen:
                wt = k.permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last).to(dev)
                def mi():
                    y = F.conv2d(x.permute(0, 3, 1, 2), wt, None, 1, 1).permute(0, 2, 3, 1)
                    return nops.bias_act_(y if y.is_contiguous() else y.contiguous(), bias, 0.1)
                t_mi = timeit(mi)
                line += f" | miopen+epilogue {t_mi:8.1f} us {fl / t_mi / 1e6:6.1f} TF/s | speedup {t_mi / t_mine:.2f}x"
                tot_o += t_mi
            print(line, flush=True)
            tot_m += t_mine; tot_f += fl
            cin = co
        print(f"L{lvl} stack: mfma {tot_m:.0f} us ({tot_f / tot_m / 1e6:.1f} TF/s)" + (f", miopen {tot_o:.0f} us" if tot_o else ""), flush=True)

if __name__ == "__main__":
    main()
