"""Parity of the HIP segment_matmul / grouped_matmul with the oracle and the golden fixtures.

Modelled on the reference's test/ops/test_matmul.py (:14-45 segment, :48-93 grouped) and
test/csrc/ops/test_matmul.cpp.  Tolerances: fp32 <= 1e-5 relative (norm-wise);
bf16/fp16 within one rounding of the correctly rounded result.
"""
import os.path as osp

import numpy as np
import pytest
import torch

import oracle
import pyg_lib_amd
from pyg_lib_amd import ops

pytestmark = pytest.mark.gpu

GOLD = np.load(osp.join(osp.dirname(__file__), 'golden', 'matmul_golden.npz'))
DEV = 'cuda:0'


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def from_bits(a):
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def test_c1_fp32_matches_golden_and_oracle():
    x, ptr, w = (torch.from_numpy(GOLD[k]) for k in ('c1_x', 'c1_ptr', 'c1_w'))
    out = ops.segment_matmul(x.to(DEV), ptr, w.to(DEV))
    assert ops.matmul_last_variant() == 'mfma_f32_k64_mc64'
    assert rel_fro(out.cpu().numpy(), GOLD['c1_out']) <= 1e-5
    assert rel_fro(out.cpu().numpy(), oracle.segment_matmul(GOLD['c1_x'], GOLD['c1_ptr'], GOLD['c1_w'])) <= 1e-5
    # atol of the reference's own test (test_matmul.py:31) scaled by sqrt(K)*sigma: elementwise check
    np.testing.assert_allclose(out.cpu().numpy(), GOLD['c1_out'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('ptr_on_device', [False, True])
def test_docstring_example_generic_path(ptr_on_device):
    x, ptr, w, b = (torch.from_numpy(GOLD[k]) for k in ('doc_x', 'doc_ptr', 'doc_w', 'doc_bias'))
    p = ptr.to(DEV) if ptr_on_device else ptr
    out = ops.segment_matmul(x.to(DEV), p, w.to(DEV))
    assert ops.matmul_last_variant() == 'mfma_f32_gen'  # K = 16 -> 32... any shape: the general-shape MFMA kernel
    np.testing.assert_allclose(out.cpu().numpy(), GOLD['doc_out'], atol=1e-5)
    outb = ops.segment_matmul(x.to(DEV), p, w.to(DEV), bias=b.to(DEV))
    np.testing.assert_allclose(outb.cpu().numpy(), GOLD['doc_out_bias'], atol=1e-5)


@pytest.mark.parametrize('ptr_on_device', [False, True])
def test_ragged_bf16_with_empty_segments(ptr_on_device):
    x, w, b = from_bits(GOLD['bf_x']), from_bits(GOLD['bf_w']), from_bits(GOLD['bf_bias'])
    ptr = torch.from_numpy(GOLD['bf_ptr'])
    p = ptr.to(DEV) if ptr_on_device else ptr
    out = ops.segment_matmul(x.to(DEV), p, w.to(DEV))
    assert ops.matmul_last_variant() == 'mfma_bf16_k128_mc128_ring'  # short relations: the item-ring kernel
    got = bits(out)
    ref = oracle.segment_matmul(GOLD['bf_x'], GOLD['bf_ptr'], GOLD['bf_w'], dtype=oracle.BF16)
    for expect in (ref, GOLD['bf_out']):
        np.testing.assert_allclose(oracle.bf16_bits_to_f32(got), oracle.bf16_bits_to_f32(expect), rtol=2 ** -7,
                                   atol=1e-6)
        assert (got == expect).mean() > 0.995
    outb = ops.segment_matmul(x.to(DEV), p, w.to(DEV), bias=b.to(DEV))
    np.testing.assert_allclose(oracle.bf16_bits_to_f32(bits(outb)), oracle.bf16_bits_to_f32(GOLD['bf_out_bias']),
                               rtol=2 ** -6, atol=1e-6)


@pytest.mark.parametrize('split', [True, False])
def test_ragged_fp32_mfma_1e5(split):
    """Both fp32 arithmetic modes (v_mfma_f32_32x32x2_f32, the default, and split-bf16 under torch's 'high' precision) against the reference's recorded
    fp32 output and the oracle: north_star's 1e-5 bar, met with two orders of magnitude to spare."""
    x = torch.from_numpy(oracle.bf16_bits_to_f32(GOLD['bf_x']))
    w = torch.from_numpy(oracle.bf16_bits_to_f32(GOLD['bf_w']))
    ptr = torch.from_numpy(GOLD['bf_ptr'])
    with ops.matmul_f32_split(split):
        out = ops.segment_matmul(x.to(DEV), ptr, w.to(DEV))
    assert ops.matmul_last_variant() in (('mfma_f32_k128_mc128_x3', 'mfma_f32_k128_regw_x3') if split else ('mfma_f32_k128_mc128',))
    assert rel_fro(out.cpu().numpy(), GOLD['f32r_out']) <= 1e-6
    assert rel_fro(out.cpu().numpy(), oracle.segment_matmul(x.numpy(), GOLD['bf_ptr'], w.numpy())) <= 1e-6


def test_fp32_split_bf16_register_w_kernel_many_short_segments():
    """Many short segments route fp32 K = M = 128 to the register-W variant of the split-bf16 kernel (W chunks and X
    tiles through one LDS-DMA ring): empty segments, 1 / 31 / 32 / 33 / 64 / 65-row segments (empty and partial second
    halves of a 64-row tile), bias, transposed weights; against float64 and against the LDS-W variant."""
    g = torch.Generator().manual_seed(12)
    rng = np.random.default_rng(12)
    sizes = [int(v) for v in rng.integers(0, 200, 400)]
    sizes[:8] = [0, 1, 31, 32, 33, 64, 65, 0]
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    for trans, with_bias in ((False, False), (False, True), (True, True)):
        x = torch.randn(n, 128, generator=g).to(DEV)
        w = torch.randn(len(sizes), 128, 128, generator=g).to(DEV)
        if trans:
            w = w.transpose(1, 2).contiguous().transpose(1, 2)
        bias = torch.randn(len(sizes), 128, generator=g).to(DEV) if with_bias else None
        with ops.matmul_f32_split(True):
            out = ops.segment_matmul(x, ptr, w, bias)
            assert ops.matmul_last_variant() == 'mfma_f32_k128_regw_x3'
            try:
                ops.set_matmul_schedule('contiguous')
                ref = ops.segment_matmul(x, ptr, w, bias)
                assert ops.matmul_last_variant() == 'mfma_f32_k128_mc128_x3'
            finally:
                ops.set_matmul_schedule('auto')
        want = torch.cat([x[int(ptr[b]):int(ptr[b + 1])].double() @ w[b].double() + (bias[b].double() if with_bias else 0)
                          for b in range(len(sizes))])
        assert (out.double() - want).norm() <= 1e-6 * want.norm()
        assert (ref.double() - want).norm() <= 1e-6 * want.norm()
        assert (out.double() - ref.double()).abs().max() <= 1e-4 * want.abs().max()  # (different k grouping per MFMA)


def test_fp32_split_bf16_is_as_accurate_as_the_fp32_mfma():
    """Full-mantissa inputs (the golden X / W above are bf16-valued, which the split represents in its first term alone),
    wide dynamic range, transposed weight views, bias, ragged segments with empty and one-row groups: the split-bf16
    result must be as close to the float64 product as the fp32 MFMA kernel's (measured 1.2e-7 vs 1.5e-7)."""
    g = torch.Generator().manual_seed(11)
    sizes = [700, 0, 1, 33, 4097, 128, 31]
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    for M, scale, trans in ((128, 1.0, False), (256, 1e4, False), (128, 1e-6, True), (384, 1.0, True)):
        x = (torch.randn(n, 128, generator=g) * scale * torch.exp(3 * torch.randn(n, 1, generator=g))).to(DEV)
        w = torch.randn(len(sizes), M, 128, generator=g).to(DEV).transpose(1, 2) if trans else \
            torch.randn(len(sizes), 128, M, generator=g).to(DEV)
        bias = torch.randn(len(sizes), M, generator=g).to(DEV) * scale
        ref = torch.cat([x[int(ptr[b]):int(ptr[b + 1])].double() @ w[b].double() + bias[b].double() for b in range(len(sizes))])
        errs = {}
        for split in (True, False):
            with ops.matmul_f32_split(split):
                out = ops.segment_matmul(x, ptr, w, bias)
            assert ops.matmul_last_variant().endswith('_x3') == split
            # row-wise: every row has its own scale
            errs[split] = ((out.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300)).max().item()
        assert errs[True] <= 1e-6 and errs[True] <= 1.5 * errs[False] + 1e-9, errs


def _special_value_case(M, seed):
    """fp32 K = 128 inputs with IEEE special values at known places; returns (x, ptr, w, rows) where `rows` names the
    affected rows.  Relation 2's weight holds an Inf, relation 3's a NaN."""
    g = torch.Generator().manual_seed(seed)
    sizes = [300, 0, 70, 33, 129, 1]
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    x = torch.randn(n, 128, generator=g)
    w = torch.randn(len(sizes), 128, M, generator=g) / 128 ** 0.5
    rows = dict(pinf=3, ninf=5, both=6, nan=7, huge=9, tiny=11, denorm=12, zero=13, infzero=14)
    x[rows['pinf'], 17] = float('inf')
    x[rows['ninf'], 100] = float('-inf')
    x[rows['both'], 2] = float('inf')
    x[rows['both'], 90] = float('-inf')
    x[rows['nan'], 64] = float('nan')
    x[rows['huge']] = 0
    x[rows['huge'], 5] = 3.3e38          # beyond the largest bf16 (3.3895e38 rounds up at 3.39e38; 3.3e38 has a bf16 hi term
    x[rows['huge'], 6] = -3.4e38         # below the limit, -3.4e38 rounds to -Inf in bf16)
    x[rows['tiny']] *= 1e-38             # around the smallest normal
    x[rows['denorm']] *= 1e-42           # fp32 denormals
    x[rows['zero']] = 0
    x[rows['infzero']] = 0
    x[rows['infzero'], 40] = float('inf')
    w[0, 40, 3] = 0.0                    # Inf * 0 = NaN in column 3 of row `infzero`
    w[0, 5] *= 1e-3                      # keep the huge products finite
    w[0, 6] *= 1e-3
    w[2, 8, 1] = float('inf')            # relation 2 (rows 300 .. 369): column 1 is +-Inf / NaN everywhere
    w[3, 0, 0] = float('nan')            # relation 3 (rows 370 .. 402): column 0 is NaN everywhere
    return x, ptr, w, rows


def _torch_cpu_reference(x, ptr, w):
    # the call the reference's CPU kernel makes per segment (ops/cpu/matmul_kernel.cpp:195-201)
    return torch.cat([x[int(ptr[b]):int(ptr[b + 1])] @ w[b] for b in range(w.size(0))])


@pytest.mark.parametrize('M', [128, 256, 64])
@pytest.mark.parametrize('sched', ['auto', 'contiguous'])
def test_fp32_special_values_exact_mode_matches_the_cpu_reference(M, sched):
    """torch's default precision ('highest') -> IEEE fp32 MFMAs: the Inf / NaN pattern (with signs) of the output equals
    the oracle's and the per-segment torch CPU matmul's, huge and tiny operands give the same finite values."""
    x, ptr, w, rows = _special_value_case(M, 77)
    assert torch.get_float32_matmul_precision() == 'highest'
    try:
        ops.set_matmul_schedule(sched)
        out = ops.segment_matmul(x.to(DEV), ptr, w.to(DEV)).cpu()
    finally:
        ops.set_matmul_schedule('auto')
    assert not ops.matmul_last_variant().endswith('_x3'), ops.matmul_last_variant()
    ref = torch.from_numpy(oracle.segment_matmul(x.numpy(), ptr.numpy(), w.numpy()))
    ref_t = _torch_cpu_reference(x, ptr, w)
    for r in (ref, ref_t):
        assert torch.equal(torch.isnan(out), torch.isnan(r))
        assert torch.equal(torch.isposinf(out), torch.isposinf(r))
        assert torch.equal(torch.isneginf(out), torch.isneginf(r))
    # what the pattern must contain
    assert torch.isinf(out[rows['pinf']]).all() and torch.isinf(out[rows['ninf']]).all()
    assert not torch.isfinite(out[rows['both']]).any() and torch.isnan(out[rows['both']]).any()  # Inf - Inf where the signs meet
    assert torch.isnan(out[rows['nan']]).all()
    assert torch.isnan(out[rows['infzero'], 3]) and torch.isinf(out[rows['infzero'], 4])
    assert torch.isfinite(out[rows['huge']]).all() and out[rows['huge']].abs().max() > 1e34
    assert (out[rows['zero']] == 0).all()
    assert not torch.isfinite(out[300:370, 1]).any() and torch.isnan(out[370:403, 0]).all()
    fin = torch.isfinite(ref)
    fin[rows['tiny']] = False      # (rows around the denormal range: absolute checks below)
    fin[rows['denorm']] = False
    err = (out.double() - ref.double())[fin].abs()
    scale = (x.double().abs() @ torch.ones(128, 1, dtype=torch.double)).expand_as(ref)[fin].clamp_min(1e-45)
    assert (err / scale).max() <= 1e-5                       # every finite element, relative to its row's magnitude
    # denormal and near-denormal rows: the products are exact in fp32 unless the matrix unit flushes denormal inputs;
    # either way the absolute error stays below ten smallest-normals
    assert (out[rows['tiny']].double() - ref[rows['tiny']].double()).abs().max() <= 1e-37
    assert (out[rows['denorm']].double() - ref[rows['denorm']].double()).abs().max() <= 1e-37


@pytest.mark.parametrize('sched', ['contiguous', 'ring'])
def test_fp32_special_values_split_mode_is_the_documented_one(sched):
    """torch.set_float32_matmul_precision('high') -> split-bf16 (PYG_HIP_MM_F32_SPLIT, the counterpart of the reference's
    TF32 switch, ops/cuda/matmul_kernel.cu:158-165).  What include/pyg_hip.h promises: NaN stays NaN; an Inf operand or
    one beyond the largest bf16 makes the outputs it feeds non-finite (NaN where the exact kernel has +-Inf or a finite
    value) and touches nothing else; every other element keeps fp32 accuracy."""
    x, ptr, w, rows = _special_value_case(128, 78)
    with ops.matmul_f32_split(True):
        try:
            ops.set_matmul_schedule(sched)
            out = ops.segment_matmul(x.to(DEV), ptr, w.to(DEV)).cpu()
        finally:
            ops.set_matmul_schedule('auto')
    assert ops.matmul_last_variant() == ('mfma_f32_k128_regw_x3' if sched == 'ring' else 'mfma_f32_k128_mc128_x3')
    ref = torch.from_numpy(oracle.segment_matmul(x.numpy(), ptr.numpy(), w.numpy()))
    nonfinite_ref = ~torch.isfinite(ref)
    # non-finite in the reference -> non-finite here (never a finite number in place of Inf / NaN)
    assert (~torch.isfinite(out))[nonfinite_ref].all()
    assert torch.equal(torch.isnan(out) & torch.isnan(ref), torch.isnan(ref))
    # the only finite reference values that may turn non-finite: row `huge` (operands beyond the bf16 range)
    lost = torch.isfinite(ref) & ~torch.isfinite(out)
    lost[rows['huge']] = False
    assert not lost.any()
    fin = torch.isfinite(ref) & torch.isfinite(out)
    fin[rows['tiny']] = False      # third terms below 2^-100 are flushed: 16 instead of 24 bits, checked below
    fin[rows['denorm']] = False
    err = (out.double() - ref.double())[fin].abs()
    scale = (x.double().abs() @ torch.ones(128, 1, dtype=torch.double)).expand_as(ref)[fin].clamp_min(1e-45)
    assert (err / scale).max() <= 1e-5
    for r in ('tiny', 'denorm'):
        assert torch.isfinite(out[rows[r]]).all()
        assert (out[rows[r]].double() - ref[rows[r]].double()).abs().max() <= 1e-3 * ref[rows[r]].abs().max() + 1e-37


def test_fp32_mode_follows_torch_precision_per_call_and_per_thread():
    """The arithmetic is chosen per call from torch's switch, the schedule is a thread-local of the binding: a thread that
    asks for 'naive' does not change what another thread's calls run."""
    import threading
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5000, 128, generator=g).to(DEV)
    w = torch.randn(2, 128, 128, generator=g).to(DEV)
    ptr = torch.tensor([0, 2500, 5000])
    seen = {}

    def other_thread():
        ops.set_matmul_schedule('naive')
        ops.segment_matmul(x, ptr, w)
        seen['other'] = ops.matmul_last_variant()

    ops.segment_matmul(x, ptr, w)
    assert ops.matmul_last_variant() == 'mfma_f32_k128_mc128'
    t = threading.Thread(target=other_thread)
    t.start()
    t.join()
    assert seen['other'] == 'naive'
    ops.segment_matmul(x, ptr, w)
    assert ops.matmul_last_variant() == 'mfma_f32_k128_mc128'      # this thread's schedule is still automatic
    for prec, want in (('high', 'mfma_f32_k128_mc128_x3'), ('medium', 'mfma_f32_k128_mc128_x3'), ('highest', 'mfma_f32_k128_mc128')):
        torch.set_float32_matmul_precision(prec)
        try:
            ops.segment_matmul(x, ptr, w)
            assert ops.matmul_last_variant() == want, (prec, ops.matmul_last_variant())
        finally:
            torch.set_float32_matmul_precision('highest')
    # the backward of a forward that ran under a schedule runs under the same one (autograd thread)
    xg = x.clone().requires_grad_()
    try:
        ops.set_matmul_schedule('naive')
        out = ops.segment_matmul(xg, ptr, w)
    finally:
        ops.set_matmul_schedule('auto')
    out.sum().backward()
    ref = torch.cat([torch.ones(2500, 128, device=DEV) @ w[b].t() for b in range(2)])
    assert (xg.grad - ref).norm() <= 1e-5 * ref.norm()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('K', [32, 64, 128, 256, 512])
@pytest.mark.parametrize('M', [32, 64, 96, 128, 192, 256])
def test_mfma_shape_sweep_vs_oracle(dtype, K, M):
    torch.manual_seed(K * 1000 + M)
    sizes = [130, 0, 1, 257, 64]
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    x = torch.randn(n, K).to(dtype)
    w = (torch.randn(len(sizes), K, M) / K ** 0.5).to(dtype)
    out = ops.segment_matmul(x.to(DEV), ptr, w.to(DEV))
    assert ops.matmul_last_variant().startswith('mfma_'), ops.matmul_last_variant()
    if dtype == torch.float32:
        ref = oracle.segment_matmul(x.numpy(), ptr.numpy(), w.numpy())
        assert rel_fro(out.cpu().numpy(), ref) <= 1e-5
    elif dtype == torch.float16:
        ref = oracle.segment_matmul(x.numpy(), ptr.numpy(), w.numpy())
        np.testing.assert_allclose(out.cpu().float().numpy(), ref.astype(np.float32), rtol=2 ** -10, atol=1e-4)
    else:
        ref = oracle.segment_matmul(bits(x), ptr.numpy(), bits(w), dtype=oracle.BF16)
        np.testing.assert_allclose(oracle.bf16_bits_to_f32(bits(out)), oracle.bf16_bits_to_f32(ref), rtol=2 ** -7,
                                   atol=1e-4)
        assert (bits(out) == ref).mean() > 0.99


@pytest.mark.parametrize('dtype,K,M', [
    (torch.bfloat16, 256, 256), (torch.bfloat16, 256, 512), (torch.bfloat16, 128, 256), (torch.bfloat16, 128, 512),
    (torch.float16, 256, 256), (torch.float32, 128, 128), (torch.float32, 128, 256), (torch.float32, 128, 64),
    (torch.bfloat16, 256, 128), (torch.bfloat16, 64, 256)])
@pytest.mark.parametrize('with_bias', [False, True])
def test_many_tiles_per_workgroup_vs_oracle(dtype, K, M, with_bias):
    """~300 tiles per workgroup-sized grid would be too slow for the oracle; 150 k rows over 37 ragged segments
    still give every persistent workgroup several tiles, relation changes and partial tiles -- the pipelined
    paths (next-tile loads in flight, hand-placed waits, double-buffered accumulators, 256-column workgroups)."""
    rng = np.random.default_rng(K + M)
    sizes = rng.integers(0, 9000, 37)
    sizes[5] = 0
    sizes[11] = 1
    sizes[20] = 128 * 40  # exact tile multiple
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    g = torch.Generator().manual_seed(K * 7 + M)
    x = torch.randn(n, K, generator=g).to(dtype)
    w = (torch.randn(len(sizes), K, M, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(len(sizes), M, generator=g).to(dtype) if with_bias else None
    out = ops.segment_matmul(x.to(DEV), ptr, w.to(DEV), None if b is None else b.to(DEV))
    assert ops.matmul_last_variant().startswith('mfma_'), ops.matmul_last_variant()
    if dtype == torch.float32:
        ref = oracle.segment_matmul(x.numpy(), ptr.numpy(), w.numpy(), None if b is None else b.numpy())
        assert rel_fro(out.cpu().numpy(), ref) <= 1e-5
    elif dtype == torch.float16:
        ref = oracle.segment_matmul(x.numpy(), ptr.numpy(), w.numpy(), None if b is None else b.numpy())
        np.testing.assert_allclose(out.cpu().float().numpy(), ref.astype(np.float32), rtol=2 ** -9, atol=2e-3)
    else:
        ref = oracle.segment_matmul(bits(x), ptr.numpy(), bits(w), None if b is None else bits(b), dtype=oracle.BF16)
        got = oracle.bf16_bits_to_f32(bits(out))
        np.testing.assert_allclose(got, oracle.bf16_bits_to_f32(ref), rtol=2 ** -6, atol=2e-2)
        assert (bits(out) == ref).mean() > 0.98
    # rows of empty segments / behind the last one do not exist; every row was written
    assert torch.isfinite(out.float()).all()


SCHED_SUFFIX = {'cyclic': '_cyc', 'ticket': '_ticket', 'contiguous': '', 'ring': '_ring'}


@pytest.fixture(params=['cyclic', 'ticket', 'ring'])
def cyclic_schedule(request):
    """The schedules next to the contiguous tile ranges: banded cyclic (mfma_rows_cyc_kernel), tickets
    (mfma_rows_ticket_kernel: wave pairs, W in registers, tiles drawn from per-XCD counters) and the item ring
    (mfma_rows_k128_ring_kernel: W slices in registers, X tiles and W chunks through one LDS-DMA ring)."""
    ops.set_matmul_schedule(request.param)
    yield SCHED_SUFFIX[request.param]
    ops.set_matmul_schedule('auto')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('with_bias', [False, True])
@pytest.mark.parametrize('ptr_on_device', [False, True])
def test_cyclic_schedule_kernel_vs_oracle(cyclic_schedule, dtype, with_bias, ptr_on_device):
    """mfma_rows_cyc_kernel (256-row workgroup tiles taken cyclically, W by LDS-DMA + transposing LDS reads, one
    barrier per relation change) forced onto a ragged problem the oracle finishes quickly: empty relations, single
    rows, exact tile multiples, runs of tiny relations (several weight switches inside one workgroup's sequence,
    relations a workgroup skips entirely) and a tail shorter than one wave."""
    rng = np.random.default_rng(7 + int(with_bias))
    sizes = rng.integers(0, 6000, 61)
    sizes[[3, 4, 17]] = 0
    sizes[8] = 1
    sizes[9] = 31
    sizes[10] = 33
    sizes[25] = 256 * 9
    sizes[30:40] = rng.integers(1, 300, 10)
    sizes[60] = 5
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, 128, generator=g).to(dtype)
    w = (torch.randn(len(sizes), 128, 128, generator=g) / 128 ** 0.5).to(dtype)
    b = torch.randn(len(sizes), 128, generator=g).to(dtype) if with_bias else None
    p = ptr.to(DEV) if ptr_on_device else ptr
    out = ops.segment_matmul(x.to(DEV), p, w.to(DEV), None if b is None else b.to(DEV))
    name = 'bf16' if dtype == torch.bfloat16 else 'f16'
    assert ops.matmul_last_variant() == f'mfma_{name}_k128_mc128{cyclic_schedule}'
    if dtype == torch.float16:
        ref = oracle.segment_matmul(x.numpy(), ptr.numpy(), w.numpy(), None if b is None else b.numpy())
        np.testing.assert_allclose(out.cpu().float().numpy(), ref.astype(np.float32), rtol=2 ** -9, atol=2e-3)
    else:
        ref = oracle.segment_matmul(bits(x), ptr.numpy(), bits(w), None if b is None else bits(b), dtype=oracle.BF16)
        np.testing.assert_allclose(oracle.bf16_bits_to_f32(bits(out)), oracle.bf16_bits_to_f32(ref), rtol=2 ** -6,
                                   atol=2e-2)
        assert (bits(out) == ref).mean() > 0.98
    # the contiguous-range kernel must give the very same bits (same MFMA order per output element)
    ops.set_matmul_schedule('contiguous')
    out2 = ops.segment_matmul(x.to(DEV), p, w.to(DEV), None if b is None else b.to(DEV))
    assert ops.matmul_last_variant() == f'mfma_{name}_k128_mc128'
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))


def test_cyclic_schedule_one_relation_and_fewer_tiles_than_workgroups(cyclic_schedule):
    for rows in (1, 255, 256, 257, 40_000):
        x = torch.randn(rows, 128, device=DEV).bfloat16()
        w = (torch.randn(1, 128, 128, device=DEV) / 11).bfloat16()
        out = ops.segment_matmul(x, torch.tensor([0, rows]), w)
        assert ops.matmul_last_variant() == 'mfma_bf16_k128_mc128' + cyclic_schedule
        ref = (x.float() @ w[0].float()).bfloat16()
        torch.testing.assert_close(out.float(), ref.float(), rtol=2 ** -6, atol=2e-2)


def test_asymmetric_weight_detects_transposes():
    # A = I picks rows of W; an asymmetric W catches row/col swaps in the MFMA C-layout.
    K = M = 128
    x = torch.eye(K).bfloat16().repeat(3, 1)
    w = (torch.arange(K * M, dtype=torch.float32).reshape(1, K, M) % 251).bfloat16()
    out = ops.segment_matmul(x.to(DEV), torch.tensor([0, 3 * K]), w.to(DEV))
    assert torch.equal(out.cpu(), w[0].repeat(3, 1))


@pytest.mark.parametrize('dtype', [torch.float64, torch.int32, torch.int64, torch.int16, torch.int8, torch.uint8])
def test_remaining_dtypes_generic_kernel(dtype):
    torch.manual_seed(5)
    ptr = torch.tensor([0, 3, 3, 10])
    if dtype.is_floating_point:
        x, w = torch.randn(10, 7, dtype=dtype), torch.randn(3, 7, 5, dtype=dtype)
    else:
        x = torch.randint(0, 4, (10, 7)).to(dtype)
        w = torch.randint(0, 4, (3, 7, 5)).to(dtype)
    out = ops.segment_matmul(x.to(DEV), ptr, w.to(DEV))
    assert ops.matmul_last_variant() == 'naive'
    ref = torch.cat([x[0:3] @ w[0], x[3:3] @ w[1], x[3:10] @ w[2]])
    if dtype == torch.float64:
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-12, atol=1e-12)
    else:
        assert torch.equal(out.cpu(), ref)


def test_argument_checks_raise_runtime_error():
    x = torch.randn(8, 16, device=DEV)
    w = torch.randn(2, 16, 32, device=DEV)
    with pytest.raises(RuntimeError):
        ops.segment_matmul(x, torch.tensor([0, 5, 8]), w.double())  # dtype mismatch
    with pytest.raises(RuntimeError):
        ops.segment_matmul(x, torch.tensor([0, 8]), w)  # ptr.numel() != B + 1
    with pytest.raises(RuntimeError):
        ops.segment_matmul(x, torch.tensor([0, 5, 8], dtype=torch.int32), w)  # Long expected
    with pytest.raises(RuntimeError):
        ops.segment_matmul(x, torch.tensor([0, 9, 8]), w)  # decreasing ptr (host check)
    with pytest.raises(RuntimeError):
        ops.segment_matmul(x, torch.tensor([0, 5, 8]), torch.randn(2, 15, 32, device=DEV))


def test_segment_matmul_backward():
    # test/ops/test_matmul.py:33-45: gradient shapes; plus values against plain autograd
    torch.manual_seed(0)
    x = torch.randn(8, 16, device=DEV, requires_grad=True)
    ptr = torch.tensor([0, 5, 8])
    w = torch.randn(2, 16, 32, device=DEV, requires_grad=True)
    b = torch.randn(2, 32, device=DEV, requires_grad=True)
    out = ops.segment_matmul(x, ptr, w, bias=b)
    out.mean().backward()
    assert x.grad.shape == x.shape and w.grad.shape == w.shape and b.grad.shape == b.shape
    x2 = x.detach().clone().requires_grad_()
    w2 = w.detach().clone().requires_grad_()
    ref = torch.cat([x2[0:5] @ w2[0] + b.detach()[0], x2[5:8] @ w2[1] + b.detach()[1]])
    ref.mean().backward()
    torch.testing.assert_close(out, ref, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(x.grad, x2.grad, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(w.grad, w2.grad, atol=1e-5, rtol=1e-5)


def test_grouped_matmul_golden_and_transposed():
    ins = [torch.from_numpy(GOLD[f'g_in{i}']) for i in range(3)]
    oth = [torch.from_numpy(GOLD[f'g_ot{i}']) for i in range(3)]
    outs = ops.grouped_matmul([a.to(DEV) for a in ins], [o.to(DEV) for o in oth])
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.cpu().numpy(), GOLD[f'g_out{i}'], atol=1e-4)
    # transposed (non-contiguous) others, test_matmul.py:60-62
    oth_t = [o.t().contiguous().to(DEV).t() for o in oth]
    assert not oth_t[0].is_contiguous()
    outs = ops.grouped_matmul([a.to(DEV) for a in ins], oth_t)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.cpu().numpy(), GOLD[f'g_out{i}'], atol=1e-4)
    biases = [torch.randn(o.size(-1), device=DEV) for o in oth]
    outs_b = ops.grouped_matmul([a.to(DEV) for a in ins], [o.to(DEV) for o in oth], biases)
    for o, ob, b in zip(outs, outs_b, biases):
        torch.testing.assert_close(ob, o + b)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('trans', [False, True])
def test_grouped_matmul_mfma_uniform_groups(dtype, trans):
    # C4-shaped (F=256) but small: variable N_i, uniform K=M=256 -> MFMA path
    torch.manual_seed(7)
    rows = [300, 1, 0, 129, 512]
    ins = [torch.randn(r, 256).to(dtype) for r in rows]
    oth = [(torch.randn(256, 256) / 16).to(dtype) for _ in rows]
    d_oth = [o.to(DEV) for o in oth]
    if trans:
        d_oth = [o.t().contiguous().t() for o in d_oth]
    outs = ops.grouped_matmul([a.to(DEV) for a in ins], d_oth)
    assert ops.matmul_last_variant().startswith('mfma_'), ops.matmul_last_variant()
    for a, o, out in zip(ins, oth, outs):
        if dtype == torch.float32:
            ref = oracle.matmul(a.numpy(), o.numpy())
            if a.size(0):
                assert rel_fro(out.cpu().numpy(), ref) <= 1e-5
        else:
            ref = oracle.matmul(bits(a), bits(o), dtype=oracle.BF16)
            np.testing.assert_allclose(oracle.bf16_bits_to_f32(bits(out)), oracle.bf16_bits_to_f32(ref),
                                       rtol=2 ** -7, atol=1e-4)


def test_grouped_matmul_outputs_can_be_modified_in_place():
    """The reference returns G independent tensors (ops/cpu/matmul_kernel.cpp:296-298); here they alias one pool but
    must not be tracked as views of a multi-output function -- in-place ops on an output of a differentiable call
    are legal there."""
    ins = [torch.randn(5, 16, device=DEV, requires_grad=True), torch.randn(6, 9, device=DEV, requires_grad=True)]
    oth = [torch.randn(16, 48, device=DEV, requires_grad=True), torch.randn(9, 42, device=DEV, requires_grad=True)]
    outs = ops.grouped_matmul(ins, oth)
    ref0 = (ins[0].detach() @ oth[0].detach()).relu()
    outs[0].relu_()
    outs[1] += 1.0
    torch.testing.assert_close(outs[0].detach(), ref0, atol=1e-4, rtol=1e-4)
    (outs[0].sum() + outs[1].sum()).backward()
    assert ins[0].grad is not None and oth[1].grad is not None
    mask = (ins[0].detach() @ oth[0].detach() > 0).float()
    torch.testing.assert_close(ins[0].grad, mask @ oth[0].detach().t(), atol=1e-4, rtol=1e-4)
    # ... and without autograd
    outs = ops.grouped_matmul([a.detach() for a in ins], [o.detach() for o in oth])
    outs[1].mul_(2.0)
    torch.testing.assert_close(outs[0], ins[0].detach() @ oth[0].detach(), atol=1e-4, rtol=1e-4)


def test_grouped_matmul_pool_writes_into_the_callers_buffer():
    torch.manual_seed(3)
    rows = [300, 0, 129, 512, 7]
    ins = [torch.randn(r, 256, device=DEV).bfloat16() for r in rows]
    oth = [(torch.randn(256, 256, device=DEV) / 16).bfloat16() for _ in rows]
    pool = torch.full((sum(rows) + 3, 256), 7.0, device=DEV, dtype=torch.bfloat16)
    outs = torch.ops.pyg.grouped_matmul_pool(ins, oth, pool[:sum(rows)])
    ref = ops.grouped_matmul(ins, oth)
    pos = 0
    for r, o, e in zip(rows, outs, ref):
        assert (r == 0 or o.data_ptr() == pool[pos:].data_ptr()) and o.shape == (r, 256)
        assert torch.equal(o, e)
        pos += r
    assert bool((pool[sum(rows):] == 7.0).all())
    with pytest.raises(RuntimeError):
        torch.ops.pyg.grouped_matmul_pool(ins, oth, pool)  # wrong number of rows


@pytest.mark.parametrize('sched', ['cyclic', 'ticket', 'contiguous', 'ring'])
def test_grouped_matmul_k128_both_schedules(sched):
    """grouped_matmul with uniform K = M = 128 reaches the same two kernels through the host-built tile tables."""
    torch.manual_seed(5)
    rows = [700, 1, 0, 256, 513, 90, 1024, 33]
    ins = [torch.randn(r, 128).bfloat16() for r in rows]
    oth = [(torch.randn(128, 128) / 11).bfloat16() for _ in rows]
    ops.set_matmul_schedule(sched)
    try:
        outs = ops.grouped_matmul([a.to(DEV) for a in ins], [o.to(DEV) for o in oth])
    finally:
        ops.set_matmul_schedule('auto')
    assert ops.matmul_last_variant() == 'mfma_bf16_k128_mc128' + SCHED_SUFFIX[sched]
    for a, o, out in zip(ins, oth, outs):
        ref = oracle.matmul(bits(a), bits(o), dtype=oracle.BF16)
        np.testing.assert_allclose(oracle.bf16_bits_to_f32(bits(out)), oracle.bf16_bits_to_f32(ref), rtol=2 ** -7,
                                   atol=1e-4)


def test_grouped_matmul_backward_shapes():
    # test/ops/test_matmul.py:77-93
    ins = [torch.randn(5, 16, device=DEV, requires_grad=True), torch.randn(6, 9, device=DEV, requires_grad=True)]
    oth = [torch.randn(16, 48, device=DEV, requires_grad=True), torch.randn(9, 42, device=DEV, requires_grad=True)]
    outs = ops.grouped_matmul(ins, oth)
    sum(o.sum() for o in outs).backward()
    for a, o in zip(ins, oth):
        assert a.grad.shape == a.shape and o.grad.shape == o.shape
        torch.testing.assert_close(a.grad, torch.ones(a.size(0), o.size(1), device=DEV) @ o.detach().t(),
                                   atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(o.grad, a.detach().t() @ torch.ones(a.size(0), o.size(1), device=DEV),
                                   atol=1e-4, rtol=1e-4)


def test_full_size_c2_properties():
    """BASELINE config C2 (154 relations, 21,111,007 rows, F=128, bf16): size-independent checks.

    With W_b = 2^(b % 5 - 2) * P_b (a signed permutation scaled by a power of two) every output
    element is an exactly representable bf16, so the result must equal the permuted, scaled input
    bit for bit; a checksum of all outputs is compared in fp64.
    """
    B, N, F = 154, 21_111_007, 128
    g = torch.Generator(device='cpu').manual_seed(0)
    frac = torch.rand(B, generator=g)
    sizes = torch.floor(frac / frac.sum() * N).long()
    sizes[-1] += N - sizes.sum()
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    gd = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N, F, device=DEV, generator=gd, dtype=torch.float32).bfloat16()
    perm = torch.stack([torch.randperm(F, generator=g) for _ in range(B)])
    sign = (torch.randint(0, 2, (B, F), generator=g) * 2 - 1).float()
    scale = 2.0 ** (torch.arange(B) % 5 - 2).float()
    w = torch.zeros(B, F, F)
    w[torch.arange(B)[:, None], perm, torch.arange(F)[None, :]] = sign * scale[:, None]
    wd = w.bfloat16().to(DEV)
    seg = torch.repeat_interleave(torch.arange(B, device=DEV), sizes.to(DEV))
    # all three tile schedules (the automatic choice at this size is the ticket kernel)
    for mode, variant in (('auto', 'mfma_bf16_k128_mc128_ticket'), ('cyclic', 'mfma_bf16_k128_mc128_cyc'),
                          ('contiguous', 'mfma_bf16_k128_mc128'), ('ring', 'mfma_bf16_k128_mc128_ring')):
        ops.set_matmul_schedule(mode)
        try:
            out = ops.segment_matmul(x, ptr, wd)
        finally:
            ops.set_matmul_schedule('auto')
        assert ops.matmul_last_variant() == variant
        # expected: out[r, j] = sign[b, j] * scale[b] * x[r, perm[b, j]]
        total = 0.0
        bad = 0
        step = 2_000_000
        for s in range(0, N, step):
            e = min(s + step, N)
            sb = seg[s:e]
            exp = torch.gather(x[s:e].float(), 1, perm.to(DEV)[sb]) * (sign * scale[:, None]).to(DEV)[sb]
            bad += int((exp.bfloat16() != out[s:e]).sum())
            total += float(out[s:e].double().sum() - exp.double().sum())
        assert bad == 0
        assert abs(total) < 1e-6
        del out


# ---- weight gradient kernel (csrc/hip/matmul_dw.hip) ---------------------------------
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('K,M', [(128, 128), (64, 64), (128, 64), (64, 128), (256, 256), (128, 256), (256, 512)])
def test_segment_matmul_weight_gradient_kernel(dtype, K, M):
    # ragged relations incl. empty ones, sizes that are not tile multiples, one relation > many tiles
    sizes = [0, 37, 128, 129, 1000, 0, 5000, 31, 257]
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    N, B = int(ptr[-1]), len(sizes)
    g = torch.Generator().manual_seed(K * 1000 + M)
    x = torch.randn(N, K, generator=g).to(dtype)
    w = (torch.randn(B, K, M, generator=g) / K ** 0.5).to(dtype)
    gy = torch.randn(N, M, generator=g).to(dtype)
    xd = x.cuda().requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    y = ops.segment_matmul(xd, ptr, wd)
    gx, gw = torch.autograd.grad(y, [xd, wd], gy.cuda())
    # float64 reference from the stored values: dW[b] = X_b^T dY_b, dX = dY W^T
    want_w = torch.stack([x[ptr[b]:ptr[b + 1]].double().t() @ gy[ptr[b]:ptr[b + 1]].double() for b in range(B)])
    want_x = torch.cat([gy[ptr[b]:ptr[b + 1]].double() @ w[b].double().t() for b in range(B)])
    eps = {torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11, torch.float32: 1e-5}[dtype]
    scale_w = want_w.abs().max().item()
    assert (gw.double().cpu() - want_w).abs().max().item() <= eps * scale_w * 1.01 + 1e-6
    assert (gx.double().cpu() - want_x).abs().max().item() <= eps * want_x.abs().max().item() * 1.01 + 1e-6
    assert gw.shape == w.shape and gw.dtype == dtype
    # device-resident ptr takes the same path; no atomics: partial sums meet in a fixed order, the bits repeat
    y2 = ops.segment_matmul(xd, ptr.cuda(), wd)
    (gw2,) = torch.autograd.grad(y2, [wd], gy.cuda())
    assert torch.equal(gw2.view(torch.uint8), gw.view(torch.uint8))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('K,M', [(128, 128), (256, 256), (64, 128), (100, 128), (128, 47), (300, 200)])
def test_weight_gradient_is_bit_reproducible(dtype, K, M):
    """the weight gradient has no float atomics any more -- the waves of a workgroup are added through
    LDS in a fixed order, partial relations through per-workgroup fp32 slabs in workgroup order (matmul_dw_out.h) -- so, as
    for the reference's per-relation at::matmul (ops/autograd/matmul_kernel.cpp:92-107), repeated calls give the SAME
    bits, also while other streams keep the chip busy and whatever the workspace held before.  Relations that span
    many workgroups, relations inside one workgroup, empty ones; specialised and general-shape kernels."""
    rng = np.random.default_rng(K * 7 + M)
    sizes = [0, 70000, 3, 0, 129, 40000, 1, 128, 5000, 0]
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    N, B = int(ptr[-1]), len(sizes)
    g = torch.Generator(device='cuda').manual_seed(K + M)
    x = torch.randn(N, K, device='cuda', generator=g).to(dtype)
    gy = torch.randn(N, M, device='cuda', generator=g).to(dtype)
    before = ops.matmul_dw_counters()
    first = torch.ops.pyg.segment_matmul_grad_other(x, ptr, gy)
    after = ops.matmul_dw_counters()
    fast = K in (64, 128, 256) and M % 64 == 0
    assert (after[0] - before[0], after[1] - before[1]) == ((1, 0) if fast else (0, 1))
    want = torch.stack([x[ptr[b]:ptr[b + 1]].double().t() @ gy[ptr[b]:ptr[b + 1]].double() for b in range(B)])
    eps = {torch.bfloat16: 2 ** -8, torch.float32: 2e-5}[dtype]
    assert (first.double() - want).abs().max().item() <= eps * want.abs().max().item() * 1.01 + 1e-6
    assert not first[0].any() and not first[3].any() and not first[9].any()   # relations without rows: zeros
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
    for rep in range(6):
        junk.random_()            # whatever the caching allocator hands the workspace next has been scribbled on
        with torch.cuda.stream(side):
            noise = torch.randn(2048, 2048, device='cuda') @ torch.randn(2048, 2048, device='cuda')
        again = torch.ops.pyg.segment_matmul_grad_other(x, ptr.cuda() if rep % 2 else ptr, gy)
        assert torch.equal(again.view(torch.uint8), first.view(torch.uint8)), rep
    torch.cuda.synchronize()
    del noise


def test_weight_gradient_is_transpose_detecting_and_linear():
    # asymmetric one-hot inputs: dW[k, m] must pick up exactly (row with x[:, k] = 1) . dY[:, m]
    K = M = 128
    ptr = torch.tensor([0, 200, 456])
    x = torch.zeros(456, K)
    gy = torch.zeros(456, M)
    x[7, 3] = 1.0
    gy[7, 100] = 2.0
    x[300, 127] = 1.0
    gy[300, 0] = -4.0
    xd = x.to(torch.bfloat16).cuda()
    wd = torch.zeros(2, K, M, dtype=torch.bfloat16, device='cuda', requires_grad=True)
    y = ops.segment_matmul(xd, ptr, wd)
    (gw,) = torch.autograd.grad(y, [wd], gy.to(torch.bfloat16).cuda())
    want = torch.zeros(2, K, M)
    want[0, 3, 100] = 2.0
    want[1, 127, 0] = -4.0
    assert torch.equal(gw.float().cpu(), want)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_grouped_matmul_backward_uses_weight_gradient_kernel(dtype):
    # C4-shaped (F=256) and F=128 groups of ragged sizes; others_grad goes through grouped_matmul(X_i^T, dY_i)
    for F in (128, 256):
        sizes = [300, 1, 4097, 128, 999]
        g = torch.Generator().manual_seed(F)
        xs = [torch.randn(n, F, generator=g).to(dtype) for n in sizes]
        ws = [(torch.randn(F, F, generator=g) / F ** 0.5).to(dtype) for _ in sizes]
        gs = [torch.randn(n, F, generator=g).to(dtype) for n in sizes]
        xd = [x.cuda().requires_grad_(True) for x in xs]
        wd = [w.cuda().requires_grad_(True) for w in ws]
        outs = ops.grouped_matmul(xd, wd)
        grads = torch.autograd.grad(outs, xd + wd, [t.cuda() for t in gs])
        eps = {torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11, torch.float32: 1e-5}[dtype]
        for i, n in enumerate(sizes):
            want_x = gs[i].double() @ ws[i].double().t()
            want_w = xs[i].double().t() @ gs[i].double()
            assert (grads[i].double().cpu() - want_x).abs().max().item() <= eps * want_x.abs().max().item() * 1.01 + 1e-6
            assert (grads[len(sizes) + i].double().cpu() - want_w).abs().max().item() <= \
                eps * want_w.abs().max().item() * 1.01 + 1e-6
        # the transposed-input call itself (what the backward issues)
        direct = torch.ops.pyg.grouped_matmul([x.t() for x in xd], [t.cuda() for t in gs])
        for i in range(len(sizes)):
            assert direct[i].shape == (F, F)
            torch.testing.assert_close(direct[i].float(), grads[len(sizes) + i].float(), rtol=2e-2, atol=eps * 64)


def test_full_size_c4_properties():
    """BASELINE config C4 on one device: grouped_matmul over 512 separate (input, weight) pairs, rows log-uniform in
    [256, 65536] (the job bench_legs.leg_c4 times), F_in = F_out = 256, bf16.  W_g = 2^(g % 5 - 2) * (signed
    permutation): every output element is an exactly representable bf16, so each group's result must equal the
    permuted, scaled input bit for bit; plus the row-count bookkeeping of the output list."""
    import bench_legs
    rows = bench_legs.c4_group_rows()
    assert len(rows) == 512 and min(rows) >= 256 and max(rows) <= 65536
    F = 256
    g = torch.Generator(device='cpu').manual_seed(4)
    gd = torch.Generator(device=DEV).manual_seed(5)
    perm = torch.stack([torch.randperm(F, generator=g) for _ in rows]).to(DEV)
    sign = (torch.randint(0, 2, (len(rows), F), generator=g) * 2 - 1).float().to(DEV)
    scale = (2.0 ** (torch.arange(len(rows)) % 5 - 2).float()).to(DEV)
    xs = [torch.randn(r, F, device=DEV, generator=gd).bfloat16() for r in rows]
    ws = []
    for i in range(len(rows)):
        w = torch.zeros(F, F, device=DEV)
        w[perm[i], torch.arange(F, device=DEV)] = sign[i] * scale[i]
        ws.append(w.bfloat16())
    # the three K = 256 kernels: W in registers (default), W in LDS with 64 rows per wave ('cyclic' selects it) and with
    # 32 rows per wave ('contiguous')
    for mode, variant in (('auto', 'mfma_bf16_k256_regw'), ('cyclic', 'mfma_bf16_k256_wide256r2'),
                          ('contiguous', 'mfma_bf16_k256_wide256')):
        ops.set_matmul_schedule(mode)
        try:
            outs = ops.grouped_matmul(xs, ws)
        finally:
            ops.set_matmul_schedule('auto')
        assert ops.matmul_last_variant() == variant
        assert len(outs) == len(rows)
        bad = 0
        for i, (x, o) in enumerate(zip(xs, outs)):
            assert o.shape == (rows[i], F) and o.dtype == torch.bfloat16
            exp = (x.float()[:, perm[i]] * (sign[i] * scale[i])).bfloat16()
            bad += int((exp != o).sum())
        assert bad == 0
        del outs


def test_ticket_kernel_random_partitions_match_contiguous_bitwise():
    """Property test of the dynamic schedule: for random ragged partitions (empty relations, single rows, runs of tiny
    relations, tails of every length) the ticket kernel -- tiles drawn from counters in whatever order the waves get
    there -- must produce the very bits of the contiguous-range kernel (same MFMA order per output element)."""
    rng = np.random.default_rng(2024)
    g = torch.Generator().manual_seed(3)
    for trial in range(24):
        B = int(rng.integers(1, 200))
        kind = trial % 4
        if kind == 0:
            sizes = rng.integers(0, 3000, B)
        elif kind == 1:
            sizes = rng.integers(0, 70, B)          # many tiny relations: a refill of W for almost every tile
        elif kind == 2:
            sizes = (rng.random(B) < 0.5) * rng.integers(1, 20000, B)  # half of the relations empty
        else:
            sizes = np.array([int(rng.integers(1, 400_000))] + [int(v) for v in rng.integers(0, 65, B - 1)])
        ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
        n = int(ptr[-1])
        if n == 0:
            continue
        x = torch.randn(n, 128, generator=g).bfloat16().to(DEV)
        w = (torch.randn(B, 128, 128, generator=g) / 11).bfloat16().to(DEV)
        bias = torch.randn(B, 128, generator=g).bfloat16().to(DEV) if trial % 3 == 0 else None
        outs = {}
        for mode in ('ticket', 'contiguous', 'ring'):
            ops.set_matmul_schedule(mode)
            try:
                outs[mode] = ops.segment_matmul(x, ptr, w, bias)
                assert ops.matmul_last_variant() == 'mfma_bf16_k128_mc128' + SCHED_SUFFIX[mode]
            finally:
                ops.set_matmul_schedule('auto')
        assert torch.equal(outs['ticket'].view(torch.int16), outs['contiguous'].view(torch.int16)), (trial, B, n)
        assert torch.equal(outs['ring'].view(torch.int16), outs['contiguous'].view(torch.int16)), (trial, B, n)


def test_ticket_kernel_on_two_streams_at_once():
    """Every call owns its counters (they live in the call's workspace): two calls in flight on different streams."""
    g = torch.Generator().manual_seed(8)
    ptr = torch.tensor([0, 150_000, 150_001, 420_000])
    xs = [torch.randn(420_000, 128, generator=g).bfloat16().to(DEV) for _ in range(2)]
    ws = [(torch.randn(3, 128, 128, generator=g) / 11).bfloat16().to(DEV) for _ in range(2)]
    ref = [ops.segment_matmul(x, ptr, w) for x, w in zip(xs, ws)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    for rep in range(5):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs[i] = ops.segment_matmul(xs[i], ptr, ws[i])
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(outs[i].view(torch.int16), ref[i].view(torch.int16))
    assert ops.matmul_last_variant() == 'mfma_bf16_k128_mc128_ticket'


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_ticket_kernel_transposed_weights(dtype):
    """`other` given as a transposed view (what the dX pass of the backward issues: grad @ W^T): the ticket kernel
    reads its register fragments straight from the [M][K] storage.  Same bits as the contiguous-range kernel, and
    the oracle's product of the logical operands."""
    torch.manual_seed(12)
    rows = [700, 1, 0, 256, 513, 90, 1024, 33, 5000]
    ins = [torch.randn(r, 128).to(dtype) for r in rows]
    oth = [(torch.randn(128, 128) / 11).to(dtype) for _ in rows]
    d_in = [a.to(DEV) for a in ins]
    d_oth = [o.to(DEV).t().contiguous().t() for o in oth]  # logical [K, M], stored [M][K]
    assert not d_oth[0].is_contiguous()
    outs = {}
    name = 'bf16' if dtype == torch.bfloat16 else 'f16'
    for mode in ('ticket', 'contiguous', 'ring'):
        ops.set_matmul_schedule(mode)
        try:
            outs[mode] = ops.grouped_matmul(d_in, d_oth)
        finally:
            ops.set_matmul_schedule('auto')
        assert ops.matmul_last_variant() == f'mfma_{name}_k128_mc128' + SCHED_SUFFIX[mode]
    for r_, c_ in zip(outs['ring'], outs['contiguous']):
        assert torch.equal(r_.view(torch.int16), c_.view(torch.int16))
    for a, o, t, c in zip(ins, oth, outs['ticket'], outs['contiguous']):
        assert torch.equal(t.view(torch.int16), c.view(torch.int16))
        if dtype == torch.bfloat16:
            ref = oracle.matmul(bits(a), bits(o), dtype=oracle.BF16)
            np.testing.assert_allclose(oracle.bf16_bits_to_f32(bits(t)), oracle.bf16_bits_to_f32(ref), rtol=2 ** -7, atol=1e-4)
        else:
            ref = oracle.matmul(a.numpy(), o.numpy())
            np.testing.assert_allclose(t.cpu().float().numpy(), ref.astype(np.float32), rtol=2 ** -9, atol=2e-3)
