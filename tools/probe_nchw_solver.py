import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
for fmt in (torch.contiguous_format, torch.channels_last):
    x = torch.randn(1, 128, 192, 640, device=dev).contiguous(memory_format=fmt)
    w = torch.randn(128, 128, 3, 3, device=dev).contiguous(memory_format=fmt)
    for _ in range(5): y = F.conv2d(x, w, None, 1, 1)
    torch.cuda.synchronize()
