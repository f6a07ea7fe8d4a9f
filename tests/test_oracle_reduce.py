"""Pins the scatter / segment_coo / gather_coo / index_sort oracle against outputs of the REAL
reference CPU kernels (tests/golden/reduce_golden.npz, recorded through oracle/_ref).  CPU only."""
import numpy as np
import pytest

import oracle
from tests.golden import reduce_cases as RC

OPS = {'sum': oracle.SUM, 'mul': oracle.MUL, 'min': oracle.MIN, 'max': oracle.MAX}


def f(a, bf16):
    return oracle.bf16_bits_to_f32(a) if bf16 else a


def close(got, ref, bf16, exact=False):
    got, ref = f(got, bf16), f(ref, bf16)
    assert got.shape == ref.shape
    if exact or not np.issubdtype(ref.dtype, np.floating):
        assert np.array_equal(got, ref)
    else:
        np.testing.assert_allclose(got.astype(np.float64), ref.astype(np.float64), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('name', RC.names('scatter'))
def test_scatter_matches_reference(name):
    c = RC.case(name)
    dt = oracle.BF16 if c['bf16'] else None
    if c['op'] == 'mean':
        if c['out0'] is not None:
            pytest.skip('mean with out= is exercised on the GPU path against the golden directly')
        got = oracle.scatter_mean(c['src'], c['index'], c['dim'], None, c['dim_size'], dt)
        if c['bf16']:
            np.testing.assert_allclose(f(got, True), f(c['res'], True), rtol=2 ** -7, atol=1e-6)
        else:
            close(got, c['res'], False)
        return
    got, arg = oracle.scatter(OPS[c['op']], c['src'], c['index'], c['dim'], c['out0'], c['dim_size'], dt)
    # the restatement follows the reference's per-element rounding: bit-exact, also for bf16
    close(got, c['res'], c['bf16'], exact=True)
    if arg is not None:
        assert np.array_equal(arg, c['arg'])


@pytest.mark.parametrize('name', RC.names('coo'))
def test_segment_coo_matches_reference(name):
    c = RC.case(name)
    dt = oracle.BF16 if c['bf16'] else None
    if c['op'] == 'sum':
        got = oracle.segment_sum_coo(c['src'], c['index'], c['out0'], c['dim_size'], dt)
        close(got, c['res'], c['bf16'], exact=True)
    elif c['op'] in ('min', 'max'):
        got, arg = oracle.segment_minmax_coo(OPS[c['op']], c['src'], c['index'], c['out0'], c['dim_size'], dt)
        close(got, c['res'], c['bf16'], exact=True)
        assert np.array_equal(arg, c['arg'])
    else:
        pytest.skip('segment_mean_coo is a composite (sum / count); checked on the GPU path against the golden')


@pytest.mark.parametrize('name', RC.names('gather'))
def test_gather_coo_matches_reference(name):
    c = RC.case(name)
    got = oracle.gather_coo(c['src'], c['index'], oracle.BF16 if c['bf16'] else None)
    assert np.array_equal(got, c['res'])


@pytest.mark.parametrize('name', RC.names('sort'))
def test_index_sort_matches_reference(name):
    c = RC.case(name)
    vals, idx = oracle.index_sort(c['keys'])
    assert np.array_equal(idx, c['idx'])
    assert np.array_equal(vals, c['keys'][c['idx']])


def test_index_sort_rejects_floats_and_2d():
    with pytest.raises(RuntimeError):
        oracle.index_sort(np.zeros((2, 2), dtype=np.int64))
    with pytest.raises((RuntimeError, KeyError)):
        oracle.index_sort(np.zeros(4, dtype=np.float32))
