#!/usr/bin/env python3
"""Random-shape cross-check of the weight-gradient kernels: the interleaved kernels (default) against the phased ones
(HTD_WGRAD_X3D=0) in a child process -- weight gradients must be the same bits, bias gradients equal to fp32 rounding -- and
both against an fp64 reference on integer operands (exact).  Also the stem kernel on random image sizes.

    python tools/fuzz_wgrad.py [n_shapes] [seed]
"""
import os
import random
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shapes(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice([1, 1, 3, 3, 3, 7])
        s = rng.choice([1, 1, 1, 2]) if k != 7 else 2
        p = {1: 0, 3: 1, 7: 3}[k]
        Ci = rng.choice([4, 8, 16, 32, 36, 64, 100, 128, 256]) if k != 7 else rng.choice([4, 8])
        Co = rng.choice([36, 48, 64, 68, 128, 132, 256])
        H, W = rng.randint(1, 40), rng.randint(1, 70)
        B = rng.randint(1, 5)
        if k == 1 and rng.random() < 0.3:
            B, H, W = rng.randint(1, 3000), 1, 1          # Linear layers
        if (H + 2 * p - k) // s + 1 <= 0 or (W + 2 * p - k) // s + 1 <= 0:
            continue
        out.append((B, Ci, H, W, Co, k, s, p))
    return out


def run(cases):
    import torch
    import torch.nn.functional as F  # noqa: F401
    from htd_amd import dense
    dev = torch.device('cuda', 0)
    res = []
    for B, Ci, H, W, Co, k, s, p in cases:
        g = torch.Generator().manual_seed(B * 7 + Ci + Co + H * 13 + W)
        x = torch.randint(-3, 4, (B, Ci, H, W), generator=g).float().to(dev).contiguous(memory_format=torch.channels_last)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        gy = torch.randint(-3, 4, (B, Co, Ho, Wo), generator=g).float().to(dev).contiguous(memory_format=torch.channels_last)
        w = torch.empty(Co, Ci, k, k, device=dev).contiguous(memory_format=torch.channels_last)
        gw, gb = dense._wgrad_launch(x, gy, w, s, p, 1, True)[:2]
        ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), s, p)
        exact = bool(torch.equal(gw.double(), ref)) and bool(torch.equal(gb.double(), gy.double().sum((0, 2, 3))))
        xr = torch.randn(B, Ci, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        gr = torch.randn(B, Co, Ho, Wo, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        gw, gb = dense._wgrad_launch(xr, gr, w, s, p, 1, True)[:2]
        res.append((exact, zlib.crc32(gw.cpu().contiguous(memory_format=torch.channels_last).numpy().tobytes()),
                    [v.hex() for v in gb.cpu().double().tolist()]))
    return res


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cases = shapes(n, seed)
    if os.environ.get('FUZZ_CHILD'):
        for exact, crc, gb in run(cases):
            print(int(exact), crc, ','.join(gb))
        return
    mine = run(cases)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), str(seed)], env=dict(os.environ, FUZZ_CHILD='1', HTD_WGRAD_X3D='0'),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    other = [l.split() for l in r.stdout.splitlines() if l and l[0] in '01']
    assert len(other) == len(cases), r.stdout[-2000:]
    bad = 0
    for case, (exact, crc, gb), (e2, crc2, gb2) in zip(cases, mine, other):
        a = [float.fromhex(v) for v in gb]
        b = [float.fromhex(v) for v in gb2.split(',')]
        tol = 2e-6 * max(1e-30, max(abs(v) for v in b))
        ok = exact and e2 == '1' and crc == int(crc2) and all(abs(u - v) <= tol for u, v in zip(a, b))
        if not ok:
            bad += 1
            print('MISMATCH', case, exact, e2, crc, crc2)
    # the stem on random image sizes against fp64
    import torch
    import torch.nn.functional as F
    from htd_amd import dense
    dev = torch.device('cuda', 0)
    rng = random.Random(seed + 1)
    for _ in range(12):
        B, H, W = rng.randint(1, 3), rng.randint(1, 150), rng.randint(1, 400)
        g = torch.Generator().manual_seed(H * 1000 + W)
        x = torch.randint(-3, 4, (B, 3, H, W), generator=g).float().to(dev).contiguous(memory_format=torch.channels_last)
        w = torch.randint(-3, 4, (64, 3, 7, 7), generator=g).float().to(dev).contiguous(memory_format=torch.channels_last)
        y = dense.conv2d(x, w, None, 2, 3, 1, relu=False)
        if not torch.equal(y.double(), F.conv2d(x.double(), w.double(), None, 2, 3)):
            bad += 1
            print('STEM MISMATCH', B, H, W)
    print(f'{len(cases)} weight-gradient shapes and 12 stem sizes checked, {bad} mismatches')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
