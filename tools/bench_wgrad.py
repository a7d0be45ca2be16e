"""m4d_conv3x3_wgrad per layer of the training configuration (384x384 crops, batch 3) against MIOpen's
aten.convolution_backward (weight gradient only) -- the framework path the kernel replaced."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m4depth_amd._lib import lib, dptr, stream_ptr, check
dev = torch.device("cuda:0")
shapes = [(3, 192, 192, 64, 128, 1), (3, 192, 192, 128, 128, 1), (3, 192, 192, 128, 96, 1), (3, 192, 192, 96, 64, 1), (3, 192, 192, 64, 32, 1),
          (3, 96, 96, 122, 128, 1), (3, 96, 96, 128, 128, 1), (3, 48, 48, 128, 128, 1), (3, 24, 24, 238, 128, 1), (3, 6, 6, 470, 128, 1),
          (3, 384, 384, 16, 16, 2), (3, 192, 192, 32, 32, 2), (3, 384, 384, 3, 16, 1)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (b, h, w, cin, cout, s) in shapes:
    oh, ow = -(-h // s), -(-w // s)
    x = torch.randn(b, h, w, cin, device=dev)
    g = torch.randn(b, oh, ow, cout, device=dev)
    wgt = torch.randn(cout, cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    dw = torch.empty_strided((cout, cin, 3, 3), (9 * cin, 1, 3 * cin, cin), device=dev)
    n_ws = int(lib.m4d_conv3x3_wgrad_workspace_floats(b, h, w, cin, cout, s))
    ws = torch.empty(n_ws, device=dev)
    def mine():
        check(lib.m4d_conv3x3_wgrad(dptr(x), dptr(g), b, h, w, cin, cout, s, dptr(ws), n_ws, ctypes.c_void_p(dw.data_ptr()), stream_ptr()), "wgrad")
    ph = max((oh - 1) * s + 3 - h, 0); pw = max((ow - 1) * s + 3 - w, 0)
    xn = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    gn = g.permute(0, 3, 1, 2)
    def miopen():
        torch.ops.aten.convolution_backward(gn, xn, wgt, None, [s, s], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
    t1 = timeit(mine)
    try:
        t2 = timeit(miopen)
    except Exception as e:
        t2 = float("nan")
    fl = 2.0 * 9 * cin * cout * b * oh * ow
    print(f"b={b} {h}x{w} {cin}->{cout} s{s}: m4d {t1:8.1f} us ({fl / t1 / 1e6:6.1f} TF/s)   MIOpen {t2:8.1f} us ({fl / t2 / 1e6:6.1f} TF/s)", flush=True)
