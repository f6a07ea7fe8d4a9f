"""Parity of the HIP index_sort / scatter_* / segment_*_coo / gather_coo with the real reference's
recorded outputs (tests/golden/reduce_golden.npz) and with the oracle.

Modelled on the reference's test/ops/test_index_sort.py, test_scatter.py, test_segment_coo.py and
test_composite.py.  Integer results, min/max values, arg indices, gathers and sort permutations are
compared bit for bit; floating sums/means within the reference's own tolerances
(test_scatter.py:80-86: default assert_close for fp32/fp64, 1e-2 for fp16/bf16).
"""
import numpy as np
import pytest
import torch

import oracle
from pyg_lib_amd import ops
from tests.golden import reduce_cases as RC

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def to_t(a, bf16=False):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a).copy())
    if bf16:
        t = t.view(torch.int16).view(torch.bfloat16)
    return t


def check(got, ref_np, bf16, sum_like):
    ref = to_t(ref_np, bf16)
    got = got.cpu()
    assert got.shape == ref.shape and got.dtype == ref.dtype
    if not ref.is_floating_point() or not sum_like:
        assert torch.equal(got, ref)  # integers, min/max values: exact
    elif ref.dtype in (torch.float16, torch.bfloat16):
        # 16-bit sums round after every update in the reference and after every atomic here, in a
        # different order: allow two bf16 ulps of the largest bucket (test_scatter.py:80-86 uses a flat
        # 1e-2 on 8 elements; the recorded cases go up to 3000 x 32).
        scale = max(1.0, float(ref.float().abs().max()))
        torch.testing.assert_close(got.float(), ref.float(), atol=scale * 2 ** -7, rtol=2e-2)
    else:
        torch.testing.assert_close(got, ref)


@pytest.mark.parametrize('name', RC.names('scatter'))
def test_scatter_matches_reference(name):
    c = RC.case(name)
    src = to_t(c['src'], c['bf16']).to(DEV)
    index = to_t(c['index']).to(DEV)
    out0 = to_t(c['out0'], c['bf16'])
    out = out0.to(DEV) if out0 is not None else None
    fn = getattr(ops, 'scatter_' + c['op'])
    res = fn(src, index, c['dim'], out, c['dim_size'])
    if c['op'] in ('min', 'max'):
        check(res[0], c['res'], c['bf16'], sum_like=False)
        assert torch.equal(res[1].cpu(), to_t(c['arg']))
    else:
        check(res, c['res'], c['bf16'], sum_like=True)
    if out is not None:
        assert (res[0] if isinstance(res, tuple) else res).data_ptr() == out.data_ptr()  # `out=` is updated in place


@pytest.mark.parametrize('name', RC.names('coo'))
def test_segment_coo_matches_reference(name):
    c = RC.case(name)
    src = to_t(c['src'], c['bf16']).to(DEV)
    index = to_t(c['index']).to(DEV)
    out0 = to_t(c['out0'], c['bf16'])
    out = out0.to(DEV) if out0 is not None else None
    res = getattr(ops, f"segment_{c['op']}_coo")(src, index, out, c['dim_size'])
    if c['op'] in ('min', 'max'):
        check(res[0], c['res'], c['bf16'], sum_like=False)
        assert torch.equal(res[1].cpu(), to_t(c['arg']))
    else:
        check(res, c['res'], c['bf16'], sum_like=True)


@pytest.mark.parametrize('name', RC.names('gather'))
def test_gather_coo_matches_reference(name):
    c = RC.case(name)
    got = ops.gather_coo(to_t(c['src'], c['bf16']).to(DEV), to_t(c['index']).to(DEV))
    assert torch.equal(got.cpu(), to_t(c['res'], c['bf16']))


@pytest.mark.parametrize('name', RC.names('sort'))
def test_index_sort_matches_reference(name):
    c = RC.case(name)
    keys = to_t(c['keys']).to(DEV)
    vals, idx = ops.index_sort(keys, c['max'])
    assert idx.dtype == torch.int64 and vals.dtype == keys.dtype
    assert torch.equal(idx.cpu(), to_t(c['idx']))
    assert torch.equal(vals.cpu(), to_t(c['keys'])[to_t(c['idx'])])


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32, torch.int16, torch.uint8, torch.int8])
@pytest.mark.parametrize('n', [1, 63, 2049, 1_000_003])
def test_index_sort_equals_stable_torch_sort(dtype, n):
    # test/ops/test_index_sort.py:28-33 (values AND indices equal torch.sort(stable=True))
    g = torch.Generator().manual_seed(n)
    hi = {torch.int64: 2 ** 45, torch.int32: 2 ** 31 - 1, torch.int16: 2 ** 15 - 1, torch.uint8: 255,
          torch.int8: 127}[dtype]
    keys = torch.randint(0, hi, (n,), generator=g).to(dtype)
    ref_v, ref_i = torch.sort(keys, stable=True)
    for mx in (None, int(keys.max()), int(keys.max()) + 1000 if dtype == torch.int64 else None):
        v, i = ops.index_sort(keys.to(DEV), mx)
        assert torch.equal(v.cpu(), ref_v) and torch.equal(i.cpu(), ref_i)
    o_v, o_i = oracle.index_sort(keys.numpy())
    assert np.array_equal(o_i, ref_i.numpy())


def test_index_sort_negative_keys_and_errors():
    keys = torch.tensor([3, -1, 2, -7, 0, 2, -1], dtype=torch.int64)
    v, i = ops.index_sort(keys.to(DEV))
    rv, ri = torch.sort(keys, stable=True)
    assert torch.equal(v.cpu(), rv) and torch.equal(i.cpu(), ri)
    with pytest.raises(RuntimeError):  # test/ops/test_index_sort.py:36-39
        ops.index_sort(torch.zeros(4, 4, dtype=torch.long, device=DEV))
    with pytest.raises(RuntimeError):
        ops.index_sort(torch.zeros(4, device=DEV))


def test_index_sort_products_scale_property():
    # C3-sized column array (~1e8 keys would take a while to verify on the host: use 2e7) -- sortedness,
    # permutation and stability are checked on the device.
    n = 20_000_000
    g = torch.Generator(device=DEV).manual_seed(0)
    keys = torch.randint(0, 2_449_029, (n,), device=DEV, generator=g)
    v, i = ops.index_sort(keys, 2_449_029)
    assert bool((v[1:] >= v[:-1]).all())
    assert torch.equal(keys[i], v)
    same = v[1:] == v[:-1]
    assert bool((i[1:][same] > i[:-1][same]).all())  # stable
    assert int(torch.bincount(i, minlength=n).max()) == 1


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16, torch.float64, torch.int32,
                                   torch.int64])
@pytest.mark.parametrize('K', [1, 5, 128])
def test_scatter_sum_random_vs_oracle(dtype, K):
    torch.manual_seed(K)
    E, N = 5000, 700
    if dtype.is_floating_point:
        src = torch.randn(E, K).to(dtype)
    else:
        src = torch.randint(-50, 50, (E, K)).to(dtype)
    index = torch.randint(0, N, (E,))
    got = ops.scatter_sum(src.to(DEV), index.to(DEV), 0, None, N)
    if dtype == torch.bfloat16:
        ref = oracle.scatter(oracle.SUM, src.view(torch.int16).numpy().view(np.uint16), index.numpy(), 0, None, N,
                             oracle.BF16)[0]
        ref = to_t(ref, True)
    else:
        ref = torch.from_numpy(oracle.scatter(oracle.SUM, src.numpy(), index.numpy(), 0, None, N)[0])
    if not dtype.is_floating_point:
        assert torch.equal(got.cpu(), ref)
    elif dtype in (torch.float16, torch.bfloat16):
        scale = max(1.0, float(ref.float().abs().max()))  # a few 16-bit ulps of the largest bucket
        torch.testing.assert_close(got.cpu().float(), ref.float(), atol=scale * 2 ** -6, rtol=2e-2)
    else:
        torch.testing.assert_close(got.cpu(), ref, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize('op', ['min', 'max'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.int64, torch.int16])
def test_scatter_minmax_random_is_bit_exact(op, dtype):
    torch.manual_seed(3)
    E, N, K = 4000, 300, 7
    src = (torch.randn(E, K) * 3).round().to(dtype)  # many ties
    index = torch.randint(0, N, (E,))
    val, arg = getattr(ops, 'scatter_' + op)(src.to(DEV), index.to(DEV), 0, None, N + 5)
    code = oracle.MIN if op == 'min' else oracle.MAX
    if dtype == torch.bfloat16:
        rv, ra = oracle.scatter(code, src.view(torch.int16).numpy().view(np.uint16), index.numpy(), 0, None, N + 5,
                                oracle.BF16)
        rv = to_t(rv, True)
    else:
        rv, ra = oracle.scatter(code, src.numpy(), index.numpy(), 0, None, N + 5)
        rv = torch.from_numpy(rv)
    assert torch.equal(val.cpu(), rv)
    assert torch.equal(arg.cpu(), torch.from_numpy(ra))


def test_segment_sum_coo_sorted_runs_match_oracle():
    torch.manual_seed(5)
    E, N, K = 20000, 900, 64
    index = torch.sort(torch.randint(0, N, (E,))).values
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        src = torch.randn(E, K).to(dtype)
        got = ops.segment_sum_coo(src.to(DEV), index.to(DEV), None, N)
        if dtype == torch.bfloat16:
            ref = to_t(oracle.segment_sum_coo(src.view(torch.int16).numpy().view(np.uint16), index.numpy(), None, N,
                                              oracle.BF16), True)
        else:
            ref = torch.from_numpy(oracle.segment_sum_coo(src.numpy(), index.numpy(), None, N))
        torch.testing.assert_close(got.cpu().float(), ref.float(), atol=tol * 10, rtol=tol)
    g = ops.gather_coo(got, index.to(DEV))
    assert torch.equal(g.cpu(), got.cpu()[index])


def test_scatter_reduce_dispatchers_and_aliases():
    src = torch.randn(10, 3, device=DEV)
    index = torch.tensor([0, 1, 0, 2, 1, 2, 3, 3, 0, 1], device=DEV)
    assert ops.scatter_add is ops.scatter_sum
    for red in ('sum', 'add', 'mul', 'mean', 'min', 'max'):
        out = ops.scatter(src, index, 0, reduce=red)
        ref = torch.zeros(4, 3, device=DEV).scatter_reduce(0, index[:, None].expand_as(src), src,
                                                           {'sum': 'sum', 'add': 'sum', 'mul': 'prod', 'mean': 'mean',
                                                            'min': 'amin', 'max': 'amax'}[red], include_self=False)
        torch.testing.assert_close(out, ref)
    sidx = torch.sort(index).values
    for red in ('sum', 'mean', 'min', 'max'):
        torch.testing.assert_close(ops.segment_coo(src, sidx, reduce=red), ops.scatter(src, sidx, 0, reduce=red))
    with pytest.raises(ValueError):
        ops.scatter(src, index, 0, reduce='median')


def test_scatter_autograd_matches_reference_formulas():
    # gradcheck in fp64, as the reference's tests do
    torch.manual_seed(0)
    index = torch.tensor([0, 1, 0, 2, 1, 2], device=DEV)
    src = torch.randn(6, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    assert torch.autograd.gradcheck(lambda s: ops.scatter_sum(s, index, 0), (src,))
    assert torch.autograd.gradcheck(lambda s: ops.scatter_mean(s, index, 0), (src,))
    assert torch.autograd.gradcheck(lambda s: ops.scatter_mul(s, index, 0), (src,))
    assert torch.autograd.gradcheck(lambda s: ops.scatter_max(s, index, 0)[0], (src,))
    assert torch.autograd.gradcheck(lambda s: ops.scatter_min(s, index, 0)[0], (src,))
    sidx = torch.tensor([0, 0, 1, 1, 1, 3], device=DEV)
    assert torch.autograd.gradcheck(lambda s: ops.segment_sum_coo(s, sidx), (src,))
    assert torch.autograd.gradcheck(lambda s: ops.segment_mean_coo(s, sidx), (src,))
    assert torch.autograd.gradcheck(lambda s: ops.segment_max_coo(s, sidx)[0], (src,))
    red = torch.randn(4, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    assert torch.autograd.gradcheck(lambda s: ops.gather_coo(s, sidx), (red,))


def test_composites_match_dense_references():
    # test/ops/test_composite.py: softmax / log_softmax / std / logsumexp per group
    torch.manual_seed(1)
    src = torch.randn(12, 4, device=DEV)
    index = torch.tensor([0, 0, 1, 1, 1, 3, 3, 3, 3, 0, 1, 3], device=DEV)
    sm = ops.scatter_softmax(src, index, 0)
    lsm = ops.scatter_log_softmax(src, index, 0)
    lse = ops.scatter_logsumexp(src, index, 0)
    std = ops.scatter_std(src, index, 0)
    for g in (0, 1, 3):
        m = index == g
        torch.testing.assert_close(sm[m], torch.softmax(src[m], 0))
        torch.testing.assert_close(lsm[m], torch.log_softmax(src[m], 0), atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(lse[g], torch.logsumexp(src[m], 0), atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(std[g], src[m].std(0), atol=1e-5, rtol=1e-5)
    assert torch.equal(lse[2], torch.zeros(4, device=DEV))  # empty bucket
    with pytest.raises(ValueError):
        ops.scatter_softmax(torch.ones(3, dtype=torch.long, device=DEV), index[:3], 0)


def test_rgcn_scale_scatter_and_gather_properties():
    """R-GCN aggregation shape: E = 700k edges, K = 128, bf16.  With integer-valued
    features every partial sum is exact in bf16, so the scattered result must equal an integer
    index_add bit for bit whatever the order of the atomics."""
    E, N, K = 700_000, 120_000, 128
    g = torch.Generator(device=DEV).manual_seed(0)
    index = torch.randint(0, N, (E,), device=DEV, generator=g)
    src_i = torch.randint(-2, 3, (E, K), device=DEV, generator=g)
    out = ops.scatter_sum(src_i.to(torch.bfloat16), index, 0, None, N)
    ref = torch.zeros(N, K, dtype=torch.long, device=DEV).index_add_(0, index, src_i)
    assert int(ref.abs().max()) < 256  # exactly representable
    assert torch.equal(out.long(), ref)
    gat = ops.gather_coo(out, index)
    assert torch.equal(gat, out[index])


@pytest.mark.parametrize('dtype,K', [(torch.float32, 1), (torch.float32, 64), (torch.bfloat16, 64), (torch.float16, 8)])
def test_scatter_mean_counts_a_popular_bucket(dtype, K):
    # the bucket sizes of a 16-bit source are counted in float32 (a count above 256 is the true count, where `+= 1` in bf16
    # stops at 256 -- and no 16-bit compare-and-swap loop on ONE popular counter); index vectors of >= 4 M entries are counted
    # through their stable sort (test_scatter_mean_large_index_counts_through_the_sort)
    torch.manual_seed(11)
    E, N = 60_000, 3000
    index = torch.randint(0, N, (E,))
    index[torch.randperm(E)[:9000]] = 77
    src = torch.randint(-4, 5, (E, K)).to(dtype)
    got = ops.scatter_mean(src.to(DEV), index.to(DEV), 0, None, N)
    cnt = torch.bincount(index, minlength=N).clamp(min=1).double()
    want = torch.zeros(N, K, dtype=torch.float64).index_add_(0, index, src.double()) / cnt[:, None]
    if dtype == torch.float32:
        torch.testing.assert_close(got.cpu().double(), want, rtol=1e-6, atol=1e-6)
    elif K * 2 >= 64:   # 16-bit rows of >= 64 bytes: the sums are float32 accumulations too
        torch.testing.assert_close(got.cpu().double(), want, rtol=2 ** -7, atol=2 ** -7)
    else:               # narrow 16-bit rows add in the storage type (like the reference): leave the popular bucket out
        keep = torch.arange(N) != 77
        torch.testing.assert_close(got.cpu().double()[keep], want[keep], rtol=2 ** -5, atol=2 ** -4)
    # constant ones into a bucket of 9000+: the mean is 1 (16-bit: only if the count did not stop at 256)
    if K * src.element_size() >= 64:
        ones = torch.ones(E, K, dtype=dtype)
        m = ops.scatter_mean(ones.to(DEV), index.to(DEV), 0, None, N)
        assert float(m[77].float().min()) > 0.99 and float(m[77].float().max()) < 1.01
    # gradients still flow (count is a constant)
    x = src.float().to(DEV).requires_grad_()
    ops.scatter_mean(x, index.to(DEV), 0, None, N).sum().backward()
    torch.testing.assert_close(x.grad.cpu()[:, 0], (1.0 / cnt)[index].float(), rtol=1e-5, atol=1e-7)


def test_scatter_mean_large_index_counts_through_the_sort():
    torch.manual_seed(12)
    E, N, K = (1 << 22) + 5, 50_000, 2
    index = torch.randint(0, N, (E,), device=DEV)
    index[torch.randperm(E, device=DEV)[:E // 8]] = 123
    src = torch.randint(-4, 5, (E, K), device=DEV).float()
    got = ops.scatter_mean(src, index, 0, None, N)
    cnt = torch.bincount(index, minlength=N).clamp(min=1).double()
    want = torch.zeros(N, K, dtype=torch.float64, device=DEV).index_add_(0, index, src.double()) / cnt[:, None]
    torch.testing.assert_close(got.double(), want, rtol=1e-6, atol=1e-6)
