"""The CPU restatement of segment_*_csr / gather_csr / softmax_csr (oracle/oracle_reduce.c) against
the outputs of the REAL reference CPU kernels (tests/golden/csr_golden.npz, recorded through
oracle/_ref).  CPU only."""
import numpy as np
import pytest

import oracle
from tests.golden import csr_cases as CC

OPS = {'sum': oracle.CSR_SUM, 'mean': oracle.CSR_MEAN, 'min': oracle.CSR_MIN, 'max': oracle.CSR_MAX}


def dt(c):
    return oracle.BF16 if c['bf16'] else None


@pytest.mark.parametrize('name', CC.names('reduce'))
def test_segment_csr_matches_reference(name):
    c = CC.case(name)
    out, arg = oracle.segment_csr(OPS[c['op']], c['src'], c['indptr'], c['out0'], dt(c))
    assert out.shape == c['res'].shape
    assert np.array_equal(out.view(np.uint8), np.ascontiguousarray(c['res']).view(np.uint8)), name
    if c['arg'] is not None:
        assert np.array_equal(arg, c['arg'])


@pytest.mark.parametrize('name', CC.names('gather'))
def test_gather_csr_matches_reference(name):
    c = CC.case(name)
    out = oracle.gather_csr(c['src'], c['indptr'], c['out0'], dt(c))
    assert np.array_equal(out.view(np.uint8), np.ascontiguousarray(c['res']).view(np.uint8))


@pytest.mark.parametrize('name', CC.names('softmax'))
def test_softmax_csr_matches_reference(name):
    c = CC.case(name)
    out = oracle.softmax_csr(c['src'], c['ptr'], c['dim'])
    assert np.array_equal(out, c['res'])
    gin = oracle.softmax_csr_backward(c['res'], c['out_grad'], c['ptr'], c['dim'])
    assert np.array_equal(gin, c['in_grad'])


def test_mean_rejects_integers():
    with pytest.raises(RuntimeError):
        oracle.segment_csr(oracle.CSR_MEAN, np.arange(6), np.array([0, 3, 6]))
