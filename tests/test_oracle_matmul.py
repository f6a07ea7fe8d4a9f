"""Pins the matmul oracle and the mt19937 restatement against committed golden fixtures (CPU only).

Fixtures come from tests/golden/make_golden.py (torch CPU `@`, the expectation the reference's own
tests encode: test/ops/test_matmul.py:14-45)."""
import os.path as osp

import numpy as np
import pytest

import oracle

GOLD = np.load(osp.join(osp.dirname(__file__), 'golden', 'matmul_golden.npz'))
WORDS = np.load(osp.join(osp.dirname(__file__), 'golden', 'mt19937_words.npz'))


def rel_fro(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def test_c1_fp32():
    out = oracle.segment_matmul(GOLD['c1_x'], GOLD['c1_ptr'], GOLD['c1_w'])
    assert rel_fro(out, GOLD['c1_out']) <= 1e-6
    np.testing.assert_allclose(out, GOLD['c1_out'], rtol=1e-4, atol=1e-4)


def test_doc_example_and_bias():
    out = oracle.segment_matmul(GOLD['doc_x'], GOLD['doc_ptr'], GOLD['doc_w'])
    np.testing.assert_allclose(out, GOLD['doc_out'], atol=1e-5)  # reference test uses atol=1e-6 on its own kernel
    outb = oracle.segment_matmul(GOLD['doc_x'], GOLD['doc_ptr'], GOLD['doc_w'], bias=GOLD['doc_bias'])
    np.testing.assert_allclose(outb, GOLD['doc_out_bias'], atol=1e-5)


def test_ragged_bf16_with_empty_segments():
    out = oracle.segment_matmul(GOLD['bf_x'], GOLD['bf_ptr'], GOLD['bf_w'], dtype=oracle.BF16)
    got = oracle.bf16_bits_to_f32(out)
    ref = oracle.bf16_bits_to_f32(GOLD['bf_out'])
    # fp32-accumulate/round-once (torch) vs double-accumulate/round-once (oracle): <= 1 bf16 ulp
    np.testing.assert_allclose(got, ref, rtol=2**-7, atol=1e-6)
    assert (out == GOLD['bf_out']).mean() > 0.995
    outb = oracle.segment_matmul(GOLD['bf_x'], GOLD['bf_ptr'], GOLD['bf_w'], bias=GOLD['bf_bias'], dtype=oracle.BF16)
    np.testing.assert_allclose(oracle.bf16_bits_to_f32(outb), oracle.bf16_bits_to_f32(GOLD['bf_out_bias']),
                               rtol=2**-6, atol=1e-6)


def test_ragged_fp32():
    x = oracle.bf16_bits_to_f32(GOLD['bf_x'])
    w = oracle.bf16_bits_to_f32(GOLD['bf_w'])
    out = oracle.segment_matmul(x, GOLD['bf_ptr'], w)
    assert rel_fro(out, GOLD['f32r_out']) <= 1e-6


def test_grouped():
    for i in range(3):
        out = oracle.matmul(GOLD[f'g_in{i}'], GOLD[f'g_ot{i}'])
        np.testing.assert_allclose(out, GOLD[f'g_out{i}'], atol=1e-4)


def test_invalid_ptr_rejected():
    with pytest.raises(RuntimeError):
        oracle.segment_matmul(GOLD['doc_x'], np.array([0, 9, 8]), GOLD['doc_w'])


@pytest.mark.parametrize('seed', [0, 12345, 123456])
def test_mt19937_matches_torch_cpu_generator(seed):
    blocks = WORDS[f'seed_{seed}']  # [3, 128]: randint once, then two in-place random_ refills
    got = oracle.mt19937_words(seed, blocks.size).reshape(blocks.shape)
    assert (got == blocks).all()
