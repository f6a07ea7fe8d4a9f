"""NOT collected by a normal run (the file name does not match test_*.py).  `python -m pytest tests/_diag_demo_gpu.py` fails on
purpose -- a scatter_sum whose expectation is wrong -- to show what tests/conftest.py appends to a failing GPU test: the
last accumulating launch, the in-process atomic self-test, and the same body repeated with hardware / CAS float atomics."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_deliberately_wrong_expectation():
    from pyg_lib_amd import ops
    src = torch.ones(20000, 128, device='cuda')
    index = torch.randint(0, 700, (20000,), device='cuda')
    out = ops.scatter_sum(src, index, 0, None, 700)
    assert float(out.sum()) == 20000 * 128 + 1   # wrong on purpose
