"""A slice of tools/fuzz_matmul.py on every `-m gpu` pass: random dtypes / shapes / partitions / schedules through
segment_matmul and grouped_matmul against float64 products (12 k cases ran clean when the ring kernels landed)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('seed', [11, 12])
def test_fuzz_slice(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_matmul.py'), '8', str(seed)],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert 'fuzz_matmul:' in out.stdout
