"""Parity of the HIP segment_*_csr / gather_csr / softmax_csr with the real reference's recorded
outputs (tests/golden/csr_golden.npz) and with the oracle on larger random inputs.

Modelled on the reference's test/ops/test_segment_csr.py and test_softmax.py.  Rows are reduced in
source order in the reference's opmath, so sums and means are compared BIT for bit as well (the
lane-split path for long, few rows and the hub rows are checked separately: integer-valued data exact, floats with a
tolerance); min/max values, arg
indices and gathers are always exact.  softmax differs from glibc's expf by at most a few ulps.
"""
import numpy as np
import pytest
import torch

import oracle
from pyg_lib_amd import ops
from tests.golden import csr_cases as CC

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
OPS = {'sum': oracle.CSR_SUM, 'mean': oracle.CSR_MEAN, 'min': oracle.CSR_MIN, 'max': oracle.CSR_MAX}


def to_t(a, bf16=False):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a).copy())
    if bf16:
        t = t.view(torch.int16).view(torch.bfloat16)
    return t


def same_bits(got, ref):
    got = got.cpu().contiguous()
    assert got.shape == ref.shape and got.dtype == ref.dtype
    if got.dtype in (torch.bfloat16, torch.float16):
        return torch.equal(got.view(torch.int16), ref.view(torch.int16))
    if got.is_floating_point():
        return torch.equal(got.view(torch.int32 if got.dtype == torch.float32 else torch.int64),
                           ref.contiguous().view(torch.int32 if got.dtype == torch.float32 else torch.int64))
    return torch.equal(got, ref)


@pytest.mark.parametrize('name', CC.names('reduce'))
def test_segment_csr_matches_reference(name):
    c = CC.case(name)
    src = to_t(c['src'], c['bf16']).to(DEV)
    indptr = to_t(c['indptr']).to(DEV)
    out0 = to_t(c['out0'], c['bf16'])
    out = out0.to(DEV) if out0 is not None else None
    res = getattr(ops, f"segment_{c['op']}_csr")(src, indptr, out)
    val = res[0] if c['op'] in ('min', 'max') else res
    want = to_t(c['res'], c['bf16'])
    # floating sums of rows of more than 512 positions (hub rows: their chunks are summed by several lanes / workgroups) equal
    # the reference's sequential sum up to rounding; every other row and every other operation bit for bit
    lens = np.diff(np.asarray(c['indptr']), axis=-1)
    if c['op'] in ('sum', 'mean') and want.is_floating_point() and lens.ndim == 1 and lens.size and lens.max() > 512:
        hub = torch.from_numpy(lens > 512)
        dim = np.asarray(c['indptr']).ndim - 1
        assert dim == 0
        torch.testing.assert_close(val.cpu()[hub].float(), want[hub].float(), rtol=2e-5 if not c['bf16'] else 2 ** -7, atol=1e-4)
        assert same_bits(val.cpu()[~hub], want[~hub]), name
    else:
        assert same_bits(val, want), name
    if c['op'] in ('min', 'max'):
        assert torch.equal(res[1].cpu(), to_t(c['arg']))
    if out is not None:
        assert val.data_ptr() == out.data_ptr()  # written in place


@pytest.mark.parametrize('name', CC.names('gather'))
def test_gather_csr_matches_reference(name):
    c = CC.case(name)
    src = to_t(c['src'], c['bf16']).to(DEV)
    indptr = to_t(c['indptr']).to(DEV)
    out = to_t(c['out0'], c['bf16']).to(DEV)
    res = ops.gather_csr(src, indptr, out)
    assert same_bits(res, to_t(c['res'], c['bf16']))


def test_gather_csr_allocates_last_offset():
    indptr = torch.tensor([0, 2, 5, 5, 6], device=DEV)
    src = torch.arange(8., device=DEV).view(4, 2)
    out = ops.gather_csr(src, indptr)
    assert out.shape == (6, 2)
    assert torch.equal(out.cpu(), src.cpu()[[0, 0, 1, 1, 1, 3]])


@pytest.mark.parametrize('name', CC.names('softmax'))
def test_softmax_csr_matches_reference(name):
    c = CC.case(name)
    src = to_t(c['src']).to(DEV)
    ptr = to_t(c['ptr']).to(DEV)
    out = ops.softmax_csr(src, ptr, c['dim'])
    torch.testing.assert_close(out.cpu(), to_t(c['res']), rtol=2e-6, atol=1e-9)
    gin = torch.ops.pyg.softmax_csr_backward(to_t(c['res']).to(DEV), to_t(c['out_grad']).to(DEV), ptr, c['dim'])
    torch.testing.assert_close(gin.cpu(), to_t(c['in_grad']), rtol=1e-5, atol=1e-7)


def random_csr(rng, rows, mean_len):
    lens = rng.poisson(mean_len, rows).astype(np.int64)
    lens[rng.integers(0, rows, rows // 10)] = 0
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16, torch.int32, torch.int64, torch.float64])
@pytest.mark.parametrize('K', [1, 5, 128])
def test_random_rows_match_oracle_bit_for_bit(dtype, K):
    rng = np.random.default_rng(21)
    indptr = random_csr(rng, 3000, 12)
    E = int(indptr[-1])
    if dtype.is_floating_point:
        src_t = torch.from_numpy(rng.standard_normal((E, K)).astype(np.float32)).to(dtype)
    else:
        src_t = torch.from_numpy(rng.integers(-50, 50, (E, K))).to(dtype)
    bf16 = dtype == torch.bfloat16
    src_np = src_t.view(torch.int16).numpy().view(np.uint16) if bf16 else src_t.numpy()
    for op in ('sum', 'mean', 'min', 'max'):
        if op == 'mean' and not dtype.is_floating_point:
            continue
        want, warg = oracle.segment_csr(OPS[op], src_np, indptr, None, oracle.BF16 if bf16 else None)
        res = getattr(ops, f'segment_{op}_csr')(src_t.to(DEV), torch.from_numpy(indptr).to(DEV))
        val = res[0] if op in ('min', 'max') else res
        assert same_bits(val, to_t(want, bf16)), (op, dtype, K)
        if op in ('min', 'max'):
            assert torch.equal(res[1].cpu(), torch.from_numpy(warg))


@pytest.mark.parametrize('K', [1, 8])
def test_long_rows_use_lane_split_within_tolerance(K):
    # few, long rows: lanes split a row (sums differ by rounding only; min/max/arg stay exact)
    rng = np.random.default_rng(22)
    lens = np.array([50_000, 0, 120_000, 3, 80_000], dtype=np.int64)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    src = rng.standard_normal((int(indptr[-1]), K)).astype(np.float32)
    s = torch.from_numpy(src).to(DEV)
    ip = torch.from_numpy(indptr).to(DEV)
    want, _ = oracle.segment_csr(oracle.CSR_SUM, src, indptr)
    torch.testing.assert_close(ops.segment_sum_csr(s, ip).cpu(), torch.from_numpy(want), rtol=1e-4, atol=1e-2)
    want, _ = oracle.segment_csr(oracle.CSR_MEAN, src, indptr)
    torch.testing.assert_close(ops.segment_mean_csr(s, ip).cpu(), torch.from_numpy(want), rtol=1e-4, atol=1e-5)
    for op, code in (('min', oracle.CSR_MIN), ('max', oracle.CSR_MAX)):
        want, warg = oracle.segment_csr(code, src, indptr)
        val, arg = getattr(ops, f'segment_{op}_csr')(s, ip)
        assert torch.equal(val.cpu(), torch.from_numpy(want)) and torch.equal(arg.cpu(), torch.from_numpy(warg))
    # ties across lanes: the first position must win
    src[:] = 1.0
    val, arg = ops.segment_min_csr(torch.from_numpy(src).to(DEV), ip)
    assert torch.equal(arg.cpu()[:, 0], torch.tensor([0, int(indptr[-1]), 50_000, 170_000, 170_003]))


def hub_csr(rng, n_short=6000, hubs=(700, 5000, 40_000)):
    """Many short rows with a few hub rows between them (a power-law graph's destinations)."""
    lens = rng.integers(0, 12, n_short)
    at = np.sort(rng.choice(n_short, len(hubs), replace=False))
    lens[at] = hubs
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), at


@pytest.mark.parametrize('dtype,K', [(torch.float32, 1), (torch.float32, 5), (torch.float32, 128), (torch.float32, 1040),
                                     (torch.bfloat16, 128), (torch.bfloat16, 24), (torch.float16, 3), (torch.int64, 3),
                                     (torch.float64, 2)])
def test_hub_rows_take_a_workgroup_each(dtype, K):
    # rows of more than 512 positions are left to segment_csr_long_kernel / gather_csr_long_kernel (csr.hip): integer-valued
    # data, so every order of the additions gives the oracle's bits; many ties, so the arg of min / max must be the FIRST match
    rng = np.random.default_rng(31 + K)
    indptr, at = hub_csr(rng, hubs=(700, 5000, 40_000) if K < 1000 else (600, 3000))
    E = int(indptr[-1])
    src_t = torch.from_numpy(rng.integers(-6, 7, (E, K)).astype(np.float32)).to(dtype)
    bf16 = dtype == torch.bfloat16
    code = oracle.BF16 if bf16 else None
    src_np = src_t.view(torch.int16).numpy().view(np.uint16) if bf16 else src_t.numpy()
    ip = torch.from_numpy(indptr).to(DEV)
    for op in ('sum', 'mean', 'min', 'max'):
        if op == 'mean' and not dtype.is_floating_point:
            continue
        want, warg = oracle.segment_csr(OPS[op], src_np, indptr, None, code)
        res = getattr(ops, f'segment_{op}_csr')(src_t.to(DEV), ip)
        val = res[0] if op in ('min', 'max') else res
        if op == 'mean' and dtype != torch.float64:   # the quotient of an exact sum: the same bits unless acc_t differs
            torch.testing.assert_close(val.cpu().float(), to_t(want, bf16).float(), rtol=1e-6 if dtype == torch.float32 else 2 ** -7, atol=0)
        else:
            assert same_bits(val, to_t(want, bf16)), (op, dtype, K)
        if op in ('min', 'max'):
            assert torch.equal(res[1].cpu(), torch.from_numpy(warg))
    # into a caller's `out` (sum accumulates into it, min / max start from it)
    N = len(indptr) - 1
    base_t = torch.from_numpy(rng.integers(-3, 4, (N, K)).astype(np.float32)).to(dtype)
    base_np = base_t.view(torch.int16).numpy().view(np.uint16) if bf16 else base_t.numpy()
    for op in ('sum', 'max'):
        want, warg = oracle.segment_csr(OPS[op], src_np, indptr, base_np, code)
        res = getattr(ops, f'segment_{op}_csr')(src_t.to(DEV), ip, base_t.clone().to(DEV))
        val = res[0] if op == 'max' else res
        assert same_bits(val, to_t(want, bf16)), (op, dtype, K, 'out')
        if op == 'max':
            assert torch.equal(res[1].cpu(), torch.from_numpy(warg))
    # the mirror image
    rows_t = torch.from_numpy(rng.integers(-50, 50, (N, K)).astype(np.float32)).to(dtype)
    got = ops.gather_csr(rows_t.to(DEV), ip)
    assert same_bits(got, torch.repeat_interleave(rows_t, torch.from_numpy(np.diff(indptr)), dim=0))


@pytest.mark.parametrize('K', [1, 4])
def test_more_hubs_than_a_workgroup_lists(K):
    # the LDS-streamed kernels keep 16 hub spans per workgroup (256 / K consecutive rows): 24 hubs in a row, sums INTO an
    # existing output (a hub handled twice would be added twice), min / max, softmax
    rng = np.random.default_rng(37)
    lens = rng.integers(0, 6, 700)
    lens[100:124] = rng.integers(4200, 6000, 24)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    E, N = int(indptr[-1]), len(lens)
    src = rng.integers(-6, 7, (E, K)).astype(np.float32)
    base = rng.integers(-3, 4, (N, K)).astype(np.float32)
    s, ip = torch.from_numpy(src).to(DEV), torch.from_numpy(indptr).to(DEV)
    for op in ('sum', 'min', 'max'):
        want, warg = oracle.segment_csr(OPS[op], src, indptr, base)
        res = getattr(ops, f'segment_{op}_csr')(s, ip, torch.from_numpy(base).to(DEV))
        val = res if op == 'sum' else res[0]
        assert torch.equal(val.cpu(), torch.from_numpy(want)), op
        if op != 'sum':
            assert torch.equal(res[1].cpu(), torch.from_numpy(warg))
    x = (rng.standard_normal((E, K)) * 3).astype(np.float32)
    out = ops.softmax_csr(torch.from_numpy(x).to(DEV), ip)
    torch.testing.assert_close(out.cpu(), torch.from_numpy(oracle.softmax_csr(x, indptr)), rtol=2e-4, atol=1e-9)


def test_hub_rows_float_sums_within_tolerance_and_repeatable():
    rng = np.random.default_rng(33)
    indptr, at = hub_csr(rng)
    src = rng.standard_normal((int(indptr[-1]), 16)).astype(np.float32)
    s, ip = torch.from_numpy(src).to(DEV), torch.from_numpy(indptr).to(DEV)
    want, _ = oracle.segment_csr(oracle.CSR_SUM, src, indptr)
    got = ops.segment_sum_csr(s, ip)
    torch.testing.assert_close(got.cpu(), torch.from_numpy(want), rtol=1e-4, atol=2e-3)
    short = np.ones(len(indptr) - 1, bool)
    short[at] = False
    assert torch.equal(got.cpu()[short], torch.from_numpy(want)[short])   # short rows: source order, the oracle's bits
    assert torch.equal(got, ops.segment_sum_csr(s, ip))
    want, _ = oracle.segment_csr(oracle.CSR_MEAN, src, indptr)
    torch.testing.assert_close(ops.segment_mean_csr(s, ip).cpu(), torch.from_numpy(want), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_scatter_onto_a_hub_destination(dtype):
    # unsorted scatter_sum / min / max with wide rows sort the index and walk the rows through perm[]: a destination that
    # collects 30 000 of 200 000 edges is a hub row there
    torch.manual_seed(35)
    E, N, K = 200_000, 20_000, 64
    index = torch.randint(0, N, (E,))
    index[torch.randperm(E)[:30_000]] = 4321
    src = torch.randint(-6, 7, (E, K)).to(dtype)
    bf16 = dtype == torch.bfloat16
    src_np = src.view(torch.int16).numpy().view(np.uint16) if bf16 else src.numpy()
    code = oracle.BF16 if bf16 else None
    got = ops.scatter_sum(src.to(DEV), index.to(DEV), 0, None, N)
    # (the reference adds bf16 to bf16 position by position; the kernel keeps a float per row: exact sums of these integers)
    exact = torch.zeros(N, K).index_add_(0, index, src.float())
    assert torch.equal(got.cpu().float(), exact.to(dtype).float())
    ref = to_t(oracle.scatter(oracle.SUM, src_np, index.numpy(), 0, None, N, code)[0], bf16)
    if bf16:   # (not the hub: 30 000 bf16 additions into one bf16 stall far below the sum)
        keep = torch.arange(N) != 4321
        torch.testing.assert_close(got.cpu().float()[keep], ref.float()[keep], rtol=2e-2, atol=1.0)
    else:
        assert same_bits(got, ref)
    for op, c in (('min', oracle.MIN), ('max', oracle.MAX)):
        val, arg = getattr(ops, 'scatter_' + op)(src.to(DEV), index.to(DEV), 0, None, N)
        rv, ra = oracle.scatter(c, src_np, index.numpy(), 0, None, N, code)
        assert same_bits(val, to_t(rv, bf16)) and torch.equal(arg.cpu(), torch.from_numpy(ra))
    sidx = torch.sort(index).values
    got = ops.segment_sum_coo(src.to(DEV), sidx.to(DEV), None, N)
    assert same_bits(got, to_t(oracle.segment_sum_coo(src_np, sidx.numpy(), None, N, code), bf16))


def test_softmax_long_groups_and_oracle():
    rng = np.random.default_rng(23)
    ptr = np.array([0, 40_000, 40_001, 40_001, 100_000], dtype=np.int64)
    src = (rng.standard_normal((100_000, 2)) * 5).astype(np.float32)
    out = ops.softmax_csr(torch.from_numpy(src).to(DEV), torch.from_numpy(ptr).to(DEV), 0)
    want = oracle.softmax_csr(src, ptr, 0)
    torch.testing.assert_close(out.cpu(), torch.from_numpy(want), rtol=1e-4, atol=1e-9)
    assert torch.equal(out[40_000].cpu(), torch.ones(2))


@pytest.mark.parametrize('shape,dim', [((0, 1), 0), ((0, 3), 0), ((0, 64), 0), ((2, 0, 5), 1), ((0, 300), 0)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_softmax_hub_groups_forward_and_backward(shape, dim, dtype):
    # groups of more than 512 positions among thousands of short ones: softmax_csr_long_kernel takes them (csr.hip), for the
    # LDS-streamed kernel (inner < 16 values) and the lane kernel alike
    rng = np.random.default_rng(41 + len(shape))
    ptr, at = hub_csr(rng, n_short=3000, hubs=(600, 20_000, 3000))
    D = int(ptr[-1])
    shape = tuple(D if v == 0 else v for v in shape)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    src = (rng.standard_normal(shape) * 3).astype(npdt)
    s = torch.from_numpy(src).to(DEV).requires_grad_()
    out = ops.softmax_csr(s, torch.from_numpy(ptr).to(DEV), dim)

    def per_group(fn, *arrs):   # float64: the formulas of ops/cpu/softmax_kernel.cpp:60-222 group by group in numpy
        res = np.zeros_like(arrs[0])
        mv = [np.moveaxis(v, dim, 0) for v in arrs]
        rv = np.moveaxis(res, dim, 0)
        for g0, g1 in zip(ptr[:-1], ptr[1:]):
            if g1 > g0:
                rv[g0:g1] = fn(*[v[g0:g1] for v in mv])
        return res

    def fwd(v):
        e = np.exp(v - v.max(0, keepdims=True))
        return e / e.sum(0, keepdims=True)

    f32 = dtype == torch.float32
    want = oracle.softmax_csr(src, ptr, dim) if f32 else per_group(fwd, src)
    tol = dict(rtol=2e-4, atol=1e-9) if dtype == torch.float32 else dict(rtol=1e-11, atol=1e-300)
    torch.testing.assert_close(out.detach().cpu(), torch.from_numpy(want), **tol)
    gout = rng.standard_normal(shape).astype(npdt)
    out.backward(torch.from_numpy(gout).to(DEV))
    gwant = (oracle.softmax_csr_backward(want, gout, ptr, dim) if f32 else
             per_group(lambda o, g: o * (g - (o * g).sum(0, keepdims=True)), want, gout))
    torch.testing.assert_close(s.grad.cpu(), torch.from_numpy(gwant), rtol=2e-3 if dtype == torch.float32 else 1e-9,
                               atol=2e-6 if dtype == torch.float32 else 1e-14)
    # every group sums to one, the hubs included
    sums = ops.segment_sum_csr(out.detach().movedim(dim, 0).reshape(D, -1).contiguous(), torch.from_numpy(ptr).to(DEV))
    nonempty = torch.from_numpy(np.diff(ptr) > 0)
    torch.testing.assert_close(sums.cpu()[nonempty], torch.ones_like(sums.cpu()[nonempty]), rtol=1e-4, atol=1e-4)


def test_autograd_matches_dense_formulas():
    torch.manual_seed(0)
    indptr = torch.tensor([0, 2, 5, 5, 6], device=DEV)
    src = torch.randn(6, 3, device=DEV, dtype=torch.float64, requires_grad=True)
    w = torch.randn(4, 3, device=DEV, dtype=torch.float64)
    index = torch.tensor([0, 0, 1, 1, 1, 3], device=DEV)
    for name in ('sum', 'mean', 'min', 'max'):
        res = getattr(ops, f'segment_{name}_csr')(src, indptr)
        val = res[0] if name in ('min', 'max') else res
        (g,) = torch.autograd.grad((val * w).sum(), src)
        ref = torch.zeros(4, 3, device=DEV, dtype=torch.float64).index_reduce_(
            0, index, src, {'sum': 'mean', 'mean': 'mean', 'min': 'amin', 'max': 'amax'}[name], include_self=False)
        if name == 'sum':
            ref = torch.zeros(4, 3, device=DEV, dtype=torch.float64).index_add_(0, index, src)
        (gr,) = torch.autograd.grad((ref * w).sum(), src)
        torch.testing.assert_close(g, gr)
    ww = w.clone().requires_grad_()
    (gg,) = torch.autograd.grad((ops.gather_csr(ww, indptr) * src.detach()).sum(), ww)
    torch.testing.assert_close(gg, torch.zeros_like(w).index_add_(0, index, src.detach()))
    # softmax_csr backward against autograd through torch.softmax per group
    x = torch.randn(6, 2, device=DEV, requires_grad=True)
    ptr = torch.tensor([0, 2, 6], device=DEV)
    y = ops.softmax_csr(x, ptr, 0)
    yr = torch.cat([torch.softmax(x[:2], 0), torch.softmax(x[2:], 0)])
    c = torch.randn(6, 2, device=DEV)
    (ga,) = torch.autograd.grad((y * c).sum(), x)
    (gb,) = torch.autograd.grad((yr * c).sum(), x)
    torch.testing.assert_close(ga, gb, rtol=1e-5, atol=1e-6)


def test_errors_mirror_the_reference():
    src = torch.randn(6, 2, device=DEV)
    with pytest.raises(RuntimeError, match='same device'):
        ops.segment_sum_csr(src, torch.tensor([0, 6]))
    with pytest.raises(RuntimeError, match='src.dim'):
        ops.segment_sum_csr(torch.randn(6, device=DEV), torch.zeros(2, 3, dtype=torch.long, device=DEV))
    with pytest.raises(RuntimeError, match='not implemented'):
        ops.segment_mean_csr(torch.arange(6, device=DEV), torch.tensor([0, 6], device=DEV))
    with pytest.raises(RuntimeError, match='out.size'):
        ops.segment_sum_csr(src, torch.tensor([0, 6], device=DEV), torch.zeros(3, 2, device=DEV))
    with pytest.raises(ValueError):
        ops.segment_csr(src, torch.tensor([0, 6], device=DEV), reduce='prod')
