import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
def test(tag):
    for (cin, cout, h, w, s) in [(3, 16, 64, 128, 1), (16, 16, 64, 128, 2), (16, 32, 32, 64, 1), (64, 128, 192, 640, 1), (128, 128, 192, 640, 1), (16, 5, 192, 640, 1)]:
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = torch.randn(2, cin, h, w, device=dev).contiguous(memory_format=fmt)
            wt = torch.randn(cout, cin, 3, 3, device=dev).contiguous(memory_format=fmt)
            ys = [F.conv2d(x, wt, None, s, 1 if s == 1 else 0) for _ in range(4)]
            same = all(torch.equal(ys[0], y) for y in ys[1:])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): F.conv2d(x, wt, None, s, 1 if s == 1 else 0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            print(tag, cin, cout, h, w, s, "cl" if fmt == torch.channels_last else "nchw", "deterministic" if same else "NONDET", f"{dt*1e6:.0f} us", flush=True)
    x = torch.randn(2, 384, 1280, 16, device=dev)
    m = [x.mean(dim=(1, 2)) for _ in range(3)]
    print(tag, "mean deterministic:", all(torch.equal(m[0], y) for y in m[1:]))
test("default")
torch.backends.cudnn.deterministic = True
test("cudnn.deterministic")
torch.backends.cudnn.deterministic = False
torch.backends.cudnn.benchmark = True
test("benchmark")
