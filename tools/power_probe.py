"""Clock and power while ONE convolution shape runs back to back (is the matrix pipe power-limited?).
usage: power_probe.py Ci H W Co k [seconds] [fwd|wgrad]   -- samples `rocm-smi` from a side thread while the kernel loops."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from htd_amd import dense  # noqa: E402

Ci, H, W, Co, k = [int(v) for v in sys.argv[1:6]]
secs = float(sys.argv[6]) if len(sys.argv) > 6 else 6.0
dev = torch.device('cuda:0')
CL = torch.channels_last
x = torch.randn(4, Ci, H, W, device=dev).contiguous(memory_format=CL)
w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=CL)
mode = sys.argv[7] if len(sys.argv) > 7 else 'fwd'
dense.tag_amax(x, dense.absmax(x))          # as in the step: the input carries its maximum (H2 where the kernels have it)
y = dense._fwd_raw(x, w, None, None, 1, k // 2, 1, True)
g = torch.randn_like(y)
samples = []


def sample():
    time.sleep(1.0)
    while not done.is_set():
        r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--csv'], capture_output=True, text=True)
        samples.append(r.stdout.strip())
        time.sleep(1.0)


done = threading.Event()
th = threading.Thread(target=sample)
th.start()
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while time.time() - t0 < secs:
    e0.record()
    for _ in range(50):
        if mode == 'wgrad':
            dense._wgrad_raw(x, g, w, 1, k // 2, 1)
        else:
            dense._fwd_raw(x, w, None, None, 1, k // 2, 1, True)
    e1.record()
    torch.cuda.synchronize()
    n += 1
    last = e0.elapsed_time(e1) / 50
done.set()
th.join()
flop = 2.0 * 4 * H * W * Co * Ci * k * k
print('ms per launch %.4f  -> %.1f TFLOP/s algorithmic' % (last, flop / last / 1e9))
for s in samples[:4]:
    print(s)
