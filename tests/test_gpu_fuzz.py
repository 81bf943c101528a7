"""Random-shape exactness of the round-4 kernels on small-integer operands (tools/fuzz_r04.py): activation planes and the plane-fed
1x1 layer, the LDS-DMA bf16 convolution, the accumulating weight gradient, the RPN head gather / scatter -- each against its fp64
or tensor-formulation reference, which it must EQUAL.  (profiles/r04_fuzz_round4_kernels.log: 150 problems per kind, 0 mismatches.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('seed', [11, 12])
def test_round4_kernels_are_exact_on_random_integer_problems(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_r04.py'), '25', str(seed)], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, HTD_BF16Q_TUNE='1'))
    assert r.returncode == 0 and 'planes 0 mismatches, bf16q 0 mismatches, wacc 0 mismatches, heads 0 mismatches' in r.stdout, \
        r.stdout[-2000:] + r.stderr[-2000:]
