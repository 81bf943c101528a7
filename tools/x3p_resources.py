#!/usr/bin/env python3
"""Register / LDS / occupancy table of the conv_x3p_kernel instantiations from hipcc's kernel-resource-usage remarks.
usage: python tools/x3p_resources.py [remarks-file]   (default: compiles htd_amd/csrc/conv_x3.hip)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def remarks(extra=()):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-fhip-fp32-correctly-rounded-divide-sqrt',
           '-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(ROOT, 'htd_amd/csrc/conv_x3.hip'), '-o', '/dev/null', *extra]
    return subprocess.run(cmd, capture_output=True, text=True).stderr


def table(text):
    rows = []
    for b in re.split(r'remark: Function Name: ', text)[1:]:
        name = b.split()[0]
        m = re.search(r'conv_x3p_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)', name)
        if not m:
            continue
        g = lambda k: int(re.search(re.escape(k) + r': (\d+)', b).group(1))
        wgm, wgn, tm, tn, kw, mf16, nb = map(int, m.groups())
        rows.append((f'{wgm * tm * 32}x{wgn * tn * 32} kw{kw} {"mf16" if mf16 else "mf32"} nb{nb}', g('VGPRs'), g('AGPRs'), g('VGPRs Spill'),
                     g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]'), g('LDS Size [bytes/block]')))
    return rows


if __name__ == '__main__':
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else remarks(sys.argv[1:])
    print(f'{"kernel":28s} VGPR AGPR spill scratch occ(regs) LDS   WG/CU(LDS)')
    for r in table(text):
        print(f'{r[0]:28s} {r[1]:4d} {r[2]:4d} {r[3]:5d} {r[4]:7d} {r[5]:9d} {r[6]:6d} {163840 // r[6]:3d}')
