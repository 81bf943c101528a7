import sys, os
sys.path.insert(0, '/root/repo')
import torch
from htd_amd import capi, dense
dev = torch.device('cuda:0')
for M in (16384, 16800, 32768, 33000, 4096, 8192):
    for N, K in ((256, 2304),):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / 48
        for it in range(10):
            if it == 2: capi.profile_begin()
            dense.linear(x, w)
        prof = capi.profile_end()
        calls, ms, _, _ = prof['htd_conv2d_fwd']
        print(M, N, K, 'tiles', -(-M // 128) * (N // 128), f'{2.0 * M * N * K / (ms / calls * 1e-3) / 1e12:.1f} TF')
