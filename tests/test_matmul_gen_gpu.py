"""General-shape MFMA path of segment_matmul / grouped_matmul (csrc/hip/matmul_gen.hip) against the oracle.

The reference's grouped GEMM carries one problem size per group (pyg_lib/csrc/ops/cuda/matmul_kernel.cu:33-67) and
its own tests use K = 16 / 9 / 32 with M = 48 / 42 / 64, plain and transposed (test/ops/test_matmul.py:56-72).
Everything here must run an `mfma_*_gen` kernel, never the one-thread-per-output fallback.
Tolerances as in test_matmul_gpu.py: fp32 <= 1e-5 norm-wise, 16-bit within one rounding of the correctly rounded
result; integer-valued inputs bit-exact.
"""
import numpy as np
import pytest
import torch

import oracle
from pyg_lib_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
NAME = {torch.bfloat16: 'bf16', torch.float16: 'f16', torch.float32: 'f32'}


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def check(out, a, w, dtype, bias=None):
    """out [n, M] (device) against the oracle's a @ w (+ bias) for CPU tensors a, w of `dtype`."""
    assert out.shape == (a.size(0), w.size(1)) and out.dtype == dtype
    if a.size(0) == 0 or w.size(1) == 0:
        return
    if dtype == torch.bfloat16:
        ref = oracle.bf16_bits_to_f32(oracle.matmul(bits(a), bits(w), dtype=oracle.BF16))
        if bias is not None:  # round(product) + bias, rounded again (pyg_lib/ops/__init__.py:169-171)
            ref = (torch.from_numpy(ref).bfloat16() + bias).float().numpy()
        np.testing.assert_allclose(out.cpu().float().numpy(), ref, rtol=2 ** -6 if bias is not None else 2 ** -7, atol=2e-2)
    elif dtype == torch.float16:
        ref = oracle.matmul(a.numpy(), w.numpy()).astype(np.float32)
        if bias is not None:
            ref = (torch.from_numpy(ref).half() + bias).float().numpy()
        np.testing.assert_allclose(out.cpu().float().numpy(), ref, rtol=2 ** -9, atol=4e-3)
    else:
        ref = oracle.matmul(a.numpy(), w.numpy())
        if bias is not None:
            ref = ref + bias.numpy()
        assert rel_fro(out.cpu().numpy(), ref) <= 1e-5
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('K', [1, 7, 9, 16, 50, 100, 129, 200, 768])
@pytest.mark.parametrize('M', [1, 5, 42, 48, 100, 128, 130, 300])
def test_segment_matmul_general_shapes(dtype, K, M):
    torch.manual_seed(K * 1000 + M)
    sizes = [130, 0, 1, 257, 64, 127]
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    x = torch.randn(n, K).to(dtype)
    w = (torch.randn(len(sizes), K, M) / K ** 0.5).to(dtype)
    out = ops.segment_matmul(x.to(DEV), ptr, w.to(DEV))
    assert ops.matmul_last_variant() == f'mfma_{NAME[dtype]}_gen', ops.matmul_last_variant()
    for b in range(len(sizes)):
        check(out[ptr[b]:ptr[b + 1]], x[ptr[b]:ptr[b + 1]], w[b], dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('ptr_on_device', [False, True])
def test_segment_matmul_k100_many_tiles_with_bias(dtype, ptr_on_device):
    """ogbn-products features are 100 wide: enough rows for every XCD run of the blockIdx -> tile map to occur,
    ragged segments, empty ones, an exact tile multiple, bias epilogue."""
    rng = np.random.default_rng(100)
    sizes = rng.integers(0, 7000, 45)
    sizes[3] = 0
    sizes[9] = 1
    sizes[17] = 128 * 33
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 100, generator=g).to(dtype)
    w = (torch.randn(len(sizes), 100, 128, generator=g) / 10).to(dtype)
    b = torch.randn(len(sizes), 128, generator=g).to(dtype)
    p = ptr.to(DEV) if ptr_on_device else ptr
    out = ops.segment_matmul(x.to(DEV), p, w.to(DEV), b.to(DEV))
    assert ops.matmul_last_variant() == f'mfma_{NAME[dtype]}_gen'
    assert torch.isfinite(out.float()).all()
    for s in (0, 9, 17, 30, 44):
        check(out[ptr[s]:ptr[s + 1]], x[ptr[s]:ptr[s + 1]], w[s], dtype, bias=b[s])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('trans', [False, True])
def test_grouped_matmul_reference_test_shapes(dtype, trans):
    # test/ops/test_matmul.py:56-72
    torch.manual_seed(12345)
    ins = [torch.randn(5, 16), torch.randn(6, 9), torch.randn(3, 32)]
    oth = [torch.randn(16, 48), torch.randn(9, 42), torch.randn(32, 64)]
    ins = [a.to(dtype) for a in ins]
    oth = [o.to(dtype) for o in oth]
    d_oth = [o.to(DEV) for o in oth]
    if trans:
        d_oth = [o.t().contiguous().t() for o in d_oth]
        assert not d_oth[0].is_contiguous()
    outs = ops.grouped_matmul([a.to(DEV) for a in ins], d_oth)
    assert ops.matmul_last_variant() == f'mfma_{NAME[dtype]}_gen', ops.matmul_last_variant()
    assert len(outs) == 3
    for a, o, out in zip(ins, oth, outs):
        check(out, a, o, dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('trans', [False, True])
def test_grouped_matmul_mixed_k_hetero_dict_linear(dtype, trans):
    """A HeteroDictLinear: every node type has its own feature width, one output width; plus a ragged tail group,
    an empty group and a group whose output width differs."""
    torch.manual_seed(3)
    shapes = [(700, 100, 128), (333, 128, 128), (1025, 256, 128), (130, 768, 128), (0, 64, 128), (1, 100, 128),
              (257, 40, 72)]
    ins = [torch.randn(r, k).to(dtype) for r, k, m in shapes]
    oth = [(torch.randn(k, m) / k ** 0.5).to(dtype) for r, k, m in shapes]
    d_oth = [o.to(DEV) for o in oth]
    if trans:
        d_oth = [o.t().contiguous().t() for o in d_oth]
    outs = ops.grouped_matmul([a.to(DEV) for a in ins], d_oth)
    assert ops.matmul_last_variant() == f'mfma_{NAME[dtype]}_gen', ops.matmul_last_variant()
    for a, o, out in zip(ins, oth, outs):
        check(out, a, o, dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_element_aligned_views(dtype):
    """Inputs / weights that start at odd element offsets of their storage and have odd row lengths: every access
    of the group falls back to the widest vector that still divides base and pitch (down to one element)."""
    torch.manual_seed(9)
    for K, M, off in [(9, 7, 1), (100, 128, 1), (100, 128, 4), (33, 65, 3), (64, 64, 1)]:
        rows = 300
        xs = torch.randn(off + rows * K).to(dtype)
        ws = (torch.randn(off + K * M) / K ** 0.5).to(dtype)
        xd, wd = xs.to(DEV), ws.to(DEV)
        a = xd[off:].view(rows, K)
        o = wd[off:].view(K, M)
        assert a.is_contiguous() and (a.data_ptr() % 16 != 0 or (off * a.element_size()) % 16 == 0)
        (out,) = ops.grouped_matmul([a], [o])
        assert ops.matmul_last_variant() == f'mfma_{NAME[dtype]}_gen', (K, M, off, ops.matmul_last_variant())
        check(out, xs[off:].view(rows, K), ws[off:].view(K, M), dtype)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_integer_valued_inputs_are_bit_exact(dtype):
    """Small integers: every product and partial sum is exactly representable, so any accumulation order must give
    the exact integer result -- rows, column tails, K tails, chunk boundaries and the group map all checked bit for
    bit on ~40 k rows of mixed groups."""
    rng = np.random.default_rng(11)
    shapes = [(9000, 100, 128), (5000, 768, 128), (7000, 128, 130), (3000, 65, 33), (8000, 256, 128), (6000, 17, 200),
              (129, 1, 1), (4000, 200, 260)]
    ins, oth = [], []
    for r, k, m in shapes:
        ins.append(torch.from_numpy(rng.integers(-2, 3, (r, k)).astype(np.float32)))
        oth.append(torch.from_numpy(rng.integers(-2, 3, (k, m)).astype(np.float32)))
    outs = ops.grouped_matmul([a.to(dtype).to(DEV) for a in ins], [o.to(dtype).to(DEV) for o in oth])
    assert ops.matmul_last_variant() == f'mfma_{NAME[dtype]}_gen'
    for (r, k, m), a, o, out in zip(shapes, ins, oth, outs):
        ref = (a.double() @ o.double())
        if dtype != torch.float32:
            ref = ref.to(dtype).double()  # |values| up to 4 K: beyond 256 the 16-bit result is the rounded integer
        assert torch.equal(out.cpu().double(), ref), (r, k, m)


def test_segment_matmul_backward_general_shape():
    torch.manual_seed(1)
    ptr = torch.tensor([0, 40, 40, 300, 517])
    x = torch.randn(517, 100, device=DEV, requires_grad=True)
    w = torch.randn(4, 100, 72, device=DEV, requires_grad=True)
    out = ops.segment_matmul(x, ptr, w)
    out.backward(torch.ones_like(out))
    xr = x.detach().clone().requires_grad_()
    wr = w.detach().clone().requires_grad_()
    ref = torch.cat([xr[ptr[i]:ptr[i + 1]] @ wr[i] for i in range(4)])
    ref.backward(torch.ones_like(ref))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(x.grad, xr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(w.grad, wr.grad, rtol=1e-4, atol=1e-3)


def test_zero_width_contraction_and_wide_outputs():
    # K = 0: the product is all zeros (+ bias); M > 128: several column passes over the same X tile
    x = torch.zeros(200, 0, device=DEV)
    w = torch.zeros(2, 0, 40, device=DEV)
    out = ops.segment_matmul(x, torch.tensor([0, 50, 200]), w)
    assert out.shape == (200, 40) and (out == 0).all()
    torch.manual_seed(2)
    a = torch.randn(500, 100).bfloat16()
    o = (torch.randn(100, 1000) / 10).bfloat16()
    (out,) = ops.grouped_matmul([a.to(DEV)], [o.to(DEV)])
    assert ops.matmul_last_variant() == 'mfma_bf16_gen'
    check(out, a, o, torch.bfloat16)


# ---- general-shape weight gradient (csrc/hip/matmul_dw_gen.hip; VERDICT r3 Missing 3) ------------------------------
# The reference's backward is a B-iteration at::matmul + stack loop (ops/autograd/matmul_kernel.cpp:92-107) and a Python
# loop for grouped_matmul (pyg_lib/ops/__init__.py:88-94); here every float shape is ONE launch.  Gradients are compared
# with float64 products of the stored values: fp32 <= 1e-5 norm-wise (IEEE fp32 MFMAs, fp32 atomics), 16-bit within one
# rounding of the fp32-accumulated result.

def _dw_tol(dtype, want):
    eps = {torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11, torch.float32: 2e-6}[dtype]
    return eps * want.abs().max().item() * 1.01 + 1e-6


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('K', [1, 7, 9, 16, 50, 100, 129, 200, 768])
@pytest.mark.parametrize('M', [1, 5, 42, 48, 100, 128, 130, 300])
def test_segment_matmul_backward_general_shapes(dtype, K, M):
    """The 216-shape sweep of the forward test, through autograd: dW from the general-shape kernel (one launch), dX from
    the forward kernel on W^T read in place."""
    torch.manual_seed(K * 1000 + M)
    sizes = [130, 0, 1, 257, 64, 127]
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n, B = int(ptr[-1]), len(sizes)
    x = torch.randn(n, K).to(dtype)
    w = (torch.randn(B, K, M) / K ** 0.5).to(dtype)
    gy = torch.randn(n, M).to(dtype)
    xd, wd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    before = ops.matmul_dw_counters()
    y = ops.segment_matmul(xd, ptr, wd)
    gx, gw = torch.autograd.grad(y, [xd, wd], gy.to(DEV))
    after = ops.matmul_dw_counters()
    assert after[1] == before[1] + 1 and after[0] == before[0], (before, after)   # the general-shape dW kernel, once
    want_w = torch.stack([x[ptr[b]:ptr[b + 1]].double().t() @ gy[ptr[b]:ptr[b + 1]].double() for b in range(B)])
    want_x = torch.cat([gy[ptr[b]:ptr[b + 1]].double() @ w[b].double().t() for b in range(B)])
    assert gw.shape == w.shape and gw.dtype == dtype and gx.shape == x.shape
    assert (gw.double().cpu() - want_w).abs().max().item() <= _dw_tol(dtype, want_w)
    assert (gx.double().cpu() - want_x).abs().max().item() <= _dw_tol(dtype, want_x) * (8 if dtype != torch.float32 else 4)
    if dtype == torch.float32:
        assert (gw.double().cpu() - want_w).norm() <= 1e-5 * max(want_w.norm().item(), 1e-30)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('ptr_on_device', [False, True])
def test_segment_matmul_backward_k100_many_tiles(dtype, ptr_on_device):
    """ogbn-products-shaped training step (K = 100 -> M = 128): enough rows that every workgroup owns several tiles and
    accumulator blocks are shared between workgroups (atomics), ragged relations, empty ones, an exact tile multiple."""
    rng = np.random.default_rng(101)
    sizes = rng.integers(0, 9000, 45)
    sizes[3] = 0
    sizes[9] = 1
    sizes[17] = 128 * 33
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n, B = int(ptr[-1]), len(sizes)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, 100, generator=g).to(dtype)
    w = (torch.randn(B, 100, 128, generator=g) / 10).to(dtype)
    gy = torch.randn(n, 128, generator=g).to(dtype)
    xd, wd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    before = ops.matmul_dw_counters()
    y = ops.segment_matmul(xd, ptr.to(DEV) if ptr_on_device else ptr, wd)
    (gw,) = torch.autograd.grad(y, [wd], gy.to(DEV))
    assert ops.matmul_dw_counters()[1] == before[1] + 1
    want = torch.stack([x[ptr[b]:ptr[b + 1]].double().t() @ gy[ptr[b]:ptr[b + 1]].double() for b in range(B)])
    assert (gw.double().cpu() - want).abs().max().item() <= _dw_tol(dtype, want)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_grouped_matmul_backward_mixed_shapes_is_one_launch(dtype):
    """HeteroDictLinear-style list (K in {100, 128, 256, 768, 9}, M in {128, 42, 64}): GroupedMatmul.backward's
    others_grad = grouped_matmul(X_i^T, dY_i) is recognised and served by ONE general-shape dW launch."""
    g = torch.Generator().manual_seed(13)
    shapes = [(300, 100, 128), (1, 128, 128), (4097, 256, 42), (128, 768, 64), (999, 9, 128), (0, 50, 64), (77, 128, 128)]
    xs = [torch.randn(n, k, generator=g).to(dtype) for n, k, m in shapes]
    ws = [(torch.randn(k, m, generator=g) / k ** 0.5).to(dtype) for n, k, m in shapes]
    gs = [torch.randn(n, m, generator=g).to(dtype) for n, k, m in shapes]
    xd = [a.to(DEV).requires_grad_() for a in xs]
    wd = [a.to(DEV).requires_grad_() for a in ws]
    before = ops.matmul_dw_counters()
    outs = ops.grouped_matmul(xd, wd)
    grads = torch.autograd.grad(outs, xd + wd, [a.to(DEV) for a in gs])
    after = ops.matmul_dw_counters()
    assert sum(after) == sum(before) + 1 and after[1] == before[1] + 1, (before, after)
    for i, (a, o, gy) in enumerate(zip(xs, ws, gs)):
        want_x = gy.double() @ o.double().t()
        want_w = a.double().t() @ gy.double()
        gx, gw = grads[i], grads[len(xs) + i]
        assert gx.shape == a.shape and gw.shape == o.shape and gw.dtype == dtype
        if a.size(0) == 0:   # a group without rows: zero weight gradient, empty input gradient
            assert not gw.any()
            continue
        assert (gw.double().cpu() - want_w).abs().max().item() <= _dw_tol(dtype, want_w), i
        assert (gx.double().cpu() - want_x).abs().max().item() <= _dw_tol(dtype, want_x) * 8, i


def test_weight_gradient_general_kernel_is_exact_on_integer_data_and_unaligned_views():
    """Element-aligned (not 16-byte aligned) operands: X and dY are column slices / row-offset views of larger buffers.
    Small-integer data makes every partial sum exact, so the result must equal the float64 product bit for bit --
    zero-filled tails, per-class vector loads and the block / tile decode included."""
    g = torch.Generator().manual_seed(21)
    for dtype in (torch.bfloat16, torch.float32):
        for K, M, n in ((100, 47, 3000), (130, 129, 700), (33, 257, 1500), (257, 33, 900)):
            big_x = torch.randint(-1, 2, (n + 3, K), generator=g).float()
            big_y = torch.randint(-2, 3, (n + 5, M), generator=g).float()
            x = big_x.to(dtype).to(DEV)[3:]           # starts 3 rows in: aligned to the element only when 3 K is odd
            gy = big_y.to(dtype).to(DEV)[5:]
            ptr = torch.tensor([0, 17, 17, n // 2, n])
            wd = torch.zeros(4, K, M, dtype=dtype, device=DEV, requires_grad=True)
            y = ops.segment_matmul(x, ptr, wd)
            (gw,) = torch.autograd.grad(y, [wd], gy)
            want = torch.stack([big_x[3:][ptr[b]:ptr[b + 1]].double().t() @ big_y[5:][ptr[b]:ptr[b + 1]].double()
                                for b in range(4)])
            assert want.abs().max() < 256
            assert torch.equal(gw.double().cpu(), want), (dtype, K, M)
