"""Build libhtd_amd.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m htd_amd.csrc.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), 'libhtd_amd.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
COMMON = ['--offload-arch=' + ARCH, '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
          '-fhip-fp32-correctly-rounded-divide-sqrt'] + os.environ.get('HTD_EXTRA_HIPCC', '').split()      # experiment builds (-D...)
# per-file extra flags: NMS keeps the CPU path's unfused arithmetic (bit-exact keep sets)
EXTRA = {'nms.hip': ['-ffp-contract=off'], 'box_ops.hip': ['-ffp-contract=off'],
         'image_pipeline.hip': ['-ffp-contract=off']}


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith('.hip') or f.endswith('.cpp'))


def _newest_dep():
    deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(('.hip', '.cpp', '.h'))]
    deps.append(os.path.join(HERE, '..', '..', 'include', 'htd_amd.h'))
    return max(os.path.getmtime(d) for d in deps)


def _compile(src):
    obj = os.path.join(HERE, '_obj', src + '.o')
    hdr_time = max(os.path.getmtime(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith('.h'))
    hdr_time = max(hdr_time, os.path.getmtime(os.path.join(HERE, '..', '..', 'include', 'htd_amd.h')))
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(os.path.join(HERE, src)), hdr_time):
        return obj
    cmd = [HIPCC] + COMMON + EXTRA.get(src, []) + ['-c', os.path.join(HERE, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, ' '.join(cmd), r.stderr[-6000:]))
    return obj


def build(force=False, verbose=False):
    os.makedirs(os.path.join(HERE, '_obj'), exist_ok=True)
    if force:
        for f in os.listdir(os.path.join(HERE, '_obj')):
            os.remove(os.path.join(HERE, '_obj', f))
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) > _newest_dep():
        return LIB
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(_compile, sources()))
    cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
