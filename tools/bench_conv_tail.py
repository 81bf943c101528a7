import os, sys, torch
sys.path.insert(0, '/root/repo')
from htd_amd import dense
CL = torch.channels_last
dev = torch.device('cuda:0')
shapes = [('l3.conv2', 256, 50, 84, 256, 3), ('l4.conv2', 512, 25, 42, 512, 3), ('l3.conv1', 1024, 50, 84, 256, 1), ('l3.conv3', 256, 50, 84, 1024, 1),
          ('l4.conv3', 512, 25, 42, 2048, 1), ('l4.conv1', 2048, 25, 42, 512, 1), ('l2.conv2', 128, 100, 168, 128, 3), ('l2.conv3', 128, 100, 168, 512, 1), ('l2.conv1', 512, 100, 168, 128, 1),
          ('l1.conv2', 64, 200, 336, 64, 3), ('fpnP4', 256, 50, 84, 256, 3), ('fpnP5', 256, 25, 42, 256, 3)]
for name, Ci, H, W, Co, k in shapes:
    x = torch.randn(4, Ci, H, W, device=dev).contiguous(memory_format=CL)
    w = torch.randn(Co, Ci, k, k, device=dev).contiguous(memory_format=CL) * 0.05
    f = lambda: dense._fwd_raw(x, w, None, None, 1, k // 2, 1, True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(60): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 60
    fl = 2.0 * 4 * H * W * Co * k * k * Ci
    print(f'{name:10s} {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF')
