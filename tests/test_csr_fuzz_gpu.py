"""A slice of tools/fuzz_csr.py on every `-m gpu` pass: random row-length distributions with hub rows, row widths, dtypes,
batched offsets and given outputs through the CSR family against the oracle (exact on integer data)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('seed', [5, 6])
def test_fuzz_slice(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_csr.py'), '40', str(seed)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert f'fuzz_csr: 40 cases, seed {seed}: 0 bad' in out.stdout, out.stdout[-3000:]
