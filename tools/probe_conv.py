"""GPU probe (round 1): which activation layout does MIOpen want for the fp32
3x3 convolutions that sit on either side of the hand-written hot path?

Times the DispRefiner conv stack shapes (m4depth_network.py:103-114 channel
plan) at level 1..3 in NCHW and channels_last, b=1 and b=8, and prints device
properties.  Not part of the product; results are recorded in DESIGN.md.
"""
import os
import sys
import time
import torch
import torch.nn.functional as F


def bench(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda:0")
    p = torch.cuda.get_device_properties(0)
    print("device:", p.name, "CUs", p.multi_processor_count, "mem GB", p.total_memory / 2**30)
    print("torch", torch.__version__, "hip", torch.version.hip)
    print("env MIOPEN*/PYTORCH*:", {k: v for k, v in os.environ.items() if "MIOPEN" in k or "PYTORCH" in k})
    torch.backends.cudnn.benchmark = True
    chans = [128, 128, 96, 64, 32, 16, 5]
    for (lvl, h, w, cin) in [(1, 192, 640, 64), (2, 96, 320, 122), (3, 48, 160, 122)]:
        for b in (1, 8):
            for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
                ws = []
                c = cin
                for co in chans:
                    ws.append(torch.randn(co, c, 3, 3, device=dev).contiguous(memory_format=fmt))
                    c = co
                x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=fmt)

                def run():
                    y = x
                    for i, wt in enumerate(ws):
                        y = F.conv2d(y, wt, None, 1, 1)
                        if i < len(ws) - 1:
                            y = F.leaky_relu(y, 0.1)
                    return y

                try:
                    t = bench(run)
                    macs = 0
                    c = cin
                    for co in chans:
                        macs += 9 * c * co * h * w * b
                        c = co
                    y = run()
                    print(f"lvl{lvl} b={b} {fmt_name}: {t*1e3:8.3f} ms  {2*macs/t/1e12:6.2f} TFLOP/s  out_cl={y.is_contiguous(memory_format=torch.channels_last)}", flush=True)
                except Exception as e:  # noqa
                    print(f"lvl{lvl} b={b} {fmt_name}: FAILED {e}", flush=True)
    # per-layer at level 1, b=8, both layouts
    h, w, b = 192, 640, 8
    c = 64
    for co in chans:
        for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
            x = torch.randn(b, c, h, w, device=dev).contiguous(memory_format=fmt)
            wt = torch.randn(co, c, 3, 3, device=dev).contiguous(memory_format=fmt)
            t = bench(lambda: F.conv2d(x, wt, None, 1, 1))
            print(f"  conv {c:4d}->{co:4d} {fmt_name}: {t*1e3:8.3f} ms {2*9*c*co*h*w*b/t/1e12:6.2f} TFLOP/s", flush=True)
        c = co
    # copy bandwidth microbenchmark
    n = 1 << 28
    a = torch.empty(n, device=dev, dtype=torch.float32)
    bb = torch.empty_like(a)
    t = bench(lambda: bb.copy_(a))
    print(f"copy 1 GiB: {t*1e3:.3f} ms  {2*n*4/t/1e12:.2f} TB/s")


if __name__ == "__main__":
    main()
