"""The CPU dispatch key (csrc/binding/pyg_binding_cpu.cpp): samplers, segment/grouped_matmul and index_sort on CPU
tensors -- the operators PyG's CPU-side loaders and CPU-only users call.  Checked against the reference's own golden
vectors, the libtorch-computed biased vectors, and the oracle on random graphs (bit-exact, generator state included).
Runs without a GPU."""
import numpy as np
import pytest
import torch

import oracle
import pyg_lib_amd
from pyg_lib_amd import ops, sampler
from tests.golden import biased_cases
from tests.golden import sampler_reference_vectors as G


def t(a, dtype=torch.long):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(dtype)


@pytest.mark.parametrize('case', G.CASES, ids=[c['name'] for c in G.CASES])
def test_reference_golden_vectors_on_cpu(case):
    kw = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in case['kwargs'].items()}
    torch.manual_seed(case.get('manual_seed', 0))
    row, col, node, edge, nh, eh = sampler.neighbor_sample(t(case['rowptr']), t(case['col']), t(case['seed']),
                                                           case['num_neighbors'], **kw)
    assert row.tolist() == case['row'] and col.tolist() == case['col_out']
    assert node.tolist() == case['node'] and edge.tolist() == case['edge']
    if 'nodes_per_hop' in case:
        assert nh == case['nodes_per_hop'] and eh == case['edges_per_hop']


def test_reference_hetero_golden_vector_on_cpu():
    c = G.HETERO_CASE
    et = c['edge_types'][0]
    out = sampler.hetero_neighbor_sample({et: t(G.ROWPTR)}, {et: t(G.COL)}, {'paper': t(c['seed'])}, {et: c['num_neighbors']})
    assert out[0][et].tolist() == c['row'] and out[1][et].tolist() == c['col_out']
    assert out[2]['paper'].tolist() == c['node'] and out[3][et].tolist() == c['edge']
    assert out[4]['paper'] == c['nodes_per_hop'] and out[5][et] == c['edges_per_hop']


def random_csr(rng, n_src, n_dst, mean_deg, hub=None):
    deg = rng.poisson(mean_deg, n_src).astype(np.int64)
    deg[rng.random(n_src) < 0.1] = 0
    if hub is not None:
        deg[hub[0]] = hub[1]
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n_dst, int(rowptr[-1])).astype(np.int64)
    return rowptr, col


@pytest.mark.parametrize('replace', [False, True])
@pytest.mark.parametrize('disjoint', [False, True])
@pytest.mark.parametrize('idt', [torch.long, torch.int32])
def test_homogeneous_random_graph_matches_oracle_and_generator_state(replace, disjoint, idt):
    rng = np.random.default_rng(5 + 2 * replace + disjoint)
    rowptr, col = random_csr(rng, 3000, 3000, 9, hub=(7, 70_000))  # the hub forces 32-bit draws and engine refills
    seeds = np.concatenate([[7, 7], rng.permutation(3000)[:120]]).astype(np.int64)  # duplicate seeds included
    fan = [6, -1 if not replace else 3, 4]
    torch.manual_seed(99)
    got = sampler.neighbor_sample(t(rowptr, idt), t(col, idt), t(seeds, idt), fan, replace=replace, disjoint=disjoint)
    after = torch.randint(0, 2 ** 31, (4,)).tolist()
    ref = oracle.neighbor_sample(rowptr, col, seeds, fan, replace=replace, disjoint=disjoint, rng_seed=99)
    for g, r in zip(got[:4], ref[:4]):
        assert g.dtype == idt
        assert np.array_equal(g.numpy().astype(np.int64), np.asarray(r).reshape(g.shape))
    assert got[4] == ref[4] and got[5] == ref[5]
    # the generator ends where the reference's engine leaves it: same number of 128-word blocks drawn
    torch.manual_seed(99)
    torch.randint(-2 ** 63, 2 ** 63 - 1, (128 * ref[6]['rng_blocks'],))
    assert torch.randint(0, 2 ** 31, (4,)).tolist() == after


@pytest.mark.parametrize('csc', [False, True])
@pytest.mark.parametrize('disjoint', [False, True])
def test_hetero_random_graph_matches_oracle(csc, disjoint):
    rng = np.random.default_rng(17 + csc)
    types = ['paper', 'author', 'inst']
    sizes = {'paper': 900, 'author': 1200, 'inst': 60}
    ets = [('paper', 'cites', 'paper'), ('paper', 'rev_writes', 'author'), ('author', 'writes', 'paper'),
           ('author', 'affil', 'inst'), ('inst', 'rev_affil', 'author')]
    rp, cl = {}, {}
    for (s, r, d) in ets:
        a, b = (d, s) if csc else (s, d)
        rp[(s, r, d)], cl[(s, r, d)] = random_csr(rng, sizes[a], sizes[b], 7)
    seeds = {'paper': rng.permutation(900)[:40].astype(np.int64), 'inst': np.array([3, 5], dtype=np.int64)}
    fan = {e: [4, 3, 2] for e in ets}
    fan[ets[3]] = [-1, 0, 5]
    torch.manual_seed(4)
    got = sampler.hetero_neighbor_sample({e: t(v) for e, v in rp.items()}, {e: t(v) for e, v in cl.items()},
                                         {k: t(v) for k, v in seeds.items()}, fan, csc=csc, disjoint=disjoint)
    ref = oracle.hetero_neighbor_sample(types, ets, rp, cl, seeds, fan, csc=csc, disjoint=disjoint, rng_seed=4)
    for e in ets:
        assert np.array_equal(got[0][e].numpy(), ref[0][e]) and np.array_equal(got[1][e].numpy(), ref[1][e])
        assert np.array_equal(got[3][e].numpy(), ref[3][e]) and got[5][e] == ref[5][e]
    for ty in types:
        assert np.array_equal(got[2][ty].numpy(), np.asarray(ref[2][ty]).reshape(got[2][ty].shape)) and got[4][ty] == ref[4][ty]


@pytest.mark.parametrize('strategy', ['uniform', 'last'])
def test_temporal_sampling_matches_oracle(strategy):
    rng = np.random.default_rng(8)
    n = 500
    rowptr, col = random_csr(rng, n, n, 8)
    edge_time = np.concatenate([np.sort(rng.integers(0, 1000, rowptr[i + 1] - rowptr[i])) for i in range(n)]).astype(np.int64)
    seeds = rng.permutation(n)[:50].astype(np.int64)
    seed_time = rng.integers(200, 900, 50).astype(np.int64)
    torch.manual_seed(21)
    got = sampler.neighbor_sample(t(rowptr), t(col), t(seeds), [3, 2], edge_time=t(edge_time), seed_time=t(seed_time),
                                  disjoint=True, temporal_strategy=strategy)
    ref = oracle.neighbor_sample(rowptr, col, seeds, [3, 2], edge_time=edge_time, seed_time=seed_time, disjoint=True,
                                 temporal_strategy=strategy, rng_seed=21)
    for g, r in zip(got[:4], ref[:4]):
        assert np.array_equal(g.numpy(), np.asarray(r).reshape(g.shape))
    with pytest.raises(RuntimeError, match='non-sorted'):
        bad = edge_time.copy()
        big = int(np.argmax(np.diff(rowptr)))
        bad[rowptr[big]:rowptr[big + 1]] = bad[rowptr[big]:rowptr[big + 1]][::-1] + np.arange(rowptr[big + 1] - rowptr[big])[::-1]
        sampler.neighbor_sample(t(rowptr), t(col), t(np.array([big])), [2], edge_time=t(bad),
                                seed_time=t(np.array([10 ** 6])), disjoint=True)


BIASED = biased_cases.load()


@pytest.mark.parametrize('case', BIASED, ids=[f"c{c['id']}" for c in BIASED])
def test_biased_sampling_matches_libtorch_vectors(case):
    wd = torch.float64 if case['f64'] else torch.float32
    torch.manual_seed(case['manual_seed'])
    out = sampler.hetero_neighbor_sample({e: t(v) for e, v in case['rowptr'].items()}, {e: t(v) for e, v in case['col'].items()},
                                         {k: t(v) for k, v in case['seed'].items()}, case['fan'], disjoint=case['disjoint'],
                                         replace=case['replace'],
                                         edge_weight_dict={e: t(v, wd) for e, v in case['weight'].items()})
    for e in case['edge_types']:
        assert np.array_equal(out[0][e].numpy(), case['row_out'][e]) and np.array_equal(out[1][e].numpy(), case['col_out'][e])
        assert np.array_equal(out[3][e].numpy(), case['edge_out'][e]) and out[5][e] == case['ehops'][e]
    for ty in case['node_types']:
        assert np.array_equal(out[2][ty].numpy(), case['node'][ty].reshape(out[2][ty].shape)) and out[4][ty] == case['nhops'][ty]


def test_argument_checks_on_cpu():
    rowptr, col = t(G.ROWPTR), t(G.COL)
    with pytest.raises(RuntimeError, match='disjoint'):
        sampler.neighbor_sample(rowptr, col, t([2]), [1], node_time=torch.arange(6))
    with pytest.raises(RuntimeError, match='Undirected'):
        sampler.neighbor_sample(rowptr, col, t([2]), [1], directed=False)
    with pytest.raises(RuntimeError, match='temporal strategy'):
        sampler.neighbor_sample(rowptr, col, t([2]), [1], temporal_strategy='first')


def test_segment_and_grouped_matmul_on_cpu():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1000, 64, generator=g)
    ptr = torch.arange(0, 1001, 100)
    w = torch.randn(10, 64, 64, generator=g)
    out = ops.segment_matmul(x, ptr, w)   # BASELINE config C1 (the reference's CPU-runnable case)
    for b in range(10):
        assert torch.equal(out[ptr[b]:ptr[b + 1]], x[ptr[b]:ptr[b + 1]] @ w[b])
    bias = torch.randn(10, 64, generator=g)
    outb = ops.segment_matmul(x, ptr, w, bias)
    torch.testing.assert_close(outb[100:200], x[100:200] @ w[1] + bias[1])
    xb, wb = x.bfloat16(), w.bfloat16()
    ob = ops.segment_matmul(xb, torch.tensor([0, 0, 300, 1000]), wb[:3])
    assert torch.equal(ob[:300], xb[:300] @ wb[1]) and torch.equal(ob[300:], xb[300:] @ wb[2])
    xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
    ops.segment_matmul(xg, ptr, wg).sum().backward()
    torch.testing.assert_close(wg.grad[3], x[300:400].t() @ torch.ones(100, 64))
    torch.testing.assert_close(xg.grad[300:400], torch.ones(100, 64) @ w[3].t())
    ins = [torch.randn(5, 16, generator=g), torch.randn(0, 9, generator=g), torch.randn(6, 9, generator=g)]
    oth = [torch.randn(16, 48, generator=g), torch.randn(9, 4, generator=g), torch.randn(9, 42, generator=g)]
    for o, a, b in zip(ops.grouped_matmul(ins, oth), ins, oth):
        assert torch.equal(o, a @ b)
    with pytest.raises(RuntimeError):
        ops.segment_matmul(x, ptr.int(), w)


@pytest.mark.parametrize('dtype', [torch.int64, torch.int32, torch.int16, torch.uint8])
def test_index_sort_on_cpu(dtype):
    g = torch.Generator().manual_seed(1)
    keys = torch.randint(0, 100, (50_000,), generator=g).to(dtype)   # heavy duplicates: stability matters
    v, i = ops.index_sort(keys, 99)
    ev, ei = torch.sort(keys, stable=True)
    assert torch.equal(v, ev) and torch.equal(i, ei) and i.dtype == torch.int64
    with pytest.raises(RuntimeError):
        ops.index_sort(torch.zeros(4, 4, dtype=torch.long))
    with pytest.raises(RuntimeError):
        ops.index_sort(torch.zeros(4))


# ---- scatter / segment_coo / gather_coo / segment_csr / gather_csr / softmax_csr on CPU tensors ----------------------
# (key CPU; csrc/binding/cpu_reduce.h).  The expectations are the outputs recorded from the REAL reference build
# (tests/golden/reduce_golden.npz, csr_golden.npz; oracle/build_ref.sh): the CPU kernels keep the reference's order of
# operations, so everything -- floating sums and bf16 per-element rounding included -- must match bit for bit.

from tests.golden import reduce_cases as RC  # noqa: E402
from tests.golden import csr_cases as CC  # noqa: E402


def _t(a, bf16=False):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a).copy())
    return t.view(torch.int16).view(torch.bfloat16) if bf16 else t


def _same_bits(got, ref):
    assert got.shape == ref.shape and got.dtype == ref.dtype and got.device.type == 'cpu'
    if got.dtype in (torch.bfloat16, torch.float16):
        return torch.equal(got.contiguous().view(torch.int16), ref.contiguous().view(torch.int16))
    if got.is_floating_point():
        it = torch.int32 if got.dtype == torch.float32 else torch.int64
        return torch.equal(got.contiguous().view(it), ref.contiguous().view(it))
    return torch.equal(got, ref)


@pytest.mark.parametrize('name', RC.names('scatter'))
def test_scatter_on_cpu_matches_reference_bitwise(name):
    c = RC.case(name)
    out = _t(c['out0'], c['bf16'])
    res = getattr(ops, 'scatter_' + c['op'])(_t(c['src'], c['bf16']), _t(c['index']), c['dim'], out, c['dim_size'])
    val = res[0] if c['op'] in ('min', 'max') else res
    assert _same_bits(val, _t(c['res'], c['bf16'])), name
    if c['op'] in ('min', 'max'):
        assert torch.equal(res[1], _t(c['arg']))
    if out is not None:
        assert val.data_ptr() == out.data_ptr()


@pytest.mark.parametrize('name', RC.names('coo'))
def test_segment_coo_on_cpu_matches_reference_bitwise(name):
    c = RC.case(name)
    res = getattr(ops, f"segment_{c['op']}_coo")(_t(c['src'], c['bf16']), _t(c['index']), _t(c['out0'], c['bf16']),
                                                 c['dim_size'])
    val = res[0] if c['op'] in ('min', 'max') else res
    assert _same_bits(val, _t(c['res'], c['bf16'])), name
    if c['op'] in ('min', 'max'):
        assert torch.equal(res[1], _t(c['arg']))


@pytest.mark.parametrize('name', RC.names('gather'))
def test_gather_coo_on_cpu_matches_reference(name):
    c = RC.case(name)
    assert _same_bits(ops.gather_coo(_t(c['src'], c['bf16']), _t(c['index'])), _t(c['res'], c['bf16']))


@pytest.mark.parametrize('name', CC.names('reduce'))
def test_segment_csr_on_cpu_matches_reference_bitwise(name):
    c = CC.case(name)
    out = _t(c['out0'], c['bf16'])
    res = getattr(ops, f"segment_{c['op']}_csr")(_t(c['src'], c['bf16']), _t(c['indptr']), out)
    val = res[0] if c['op'] in ('min', 'max') else res
    assert _same_bits(val, _t(c['res'], c['bf16'])), name
    if c['op'] in ('min', 'max'):
        assert torch.equal(res[1], _t(c['arg']))
    if out is not None:
        assert val.data_ptr() == out.data_ptr()


@pytest.mark.parametrize('name', CC.names('gather'))
def test_gather_csr_on_cpu_matches_reference(name):
    c = CC.case(name)
    res = ops.gather_csr(_t(c['src'], c['bf16']), _t(c['indptr']), _t(c['out0'], c['bf16']))
    assert _same_bits(res, _t(c['res'], c['bf16']))


@pytest.mark.parametrize('name', CC.names('softmax'))
def test_softmax_csr_on_cpu_matches_reference_bitwise(name):
    c = CC.case(name)
    out = ops.softmax_csr(_t(c['src']), _t(c['ptr']), c['dim'])
    assert _same_bits(out, _t(c['res'])), name   # same libm expf, same order
    gin = torch.ops.pyg.softmax_csr_backward(_t(c['res']), _t(c['out_grad']), _t(c['ptr']), c['dim'])
    assert _same_bits(gin, _t(c['in_grad'])), name


def test_reduce_ops_on_cpu_autograd_and_composites():
    g = torch.Generator().manual_seed(0)
    src = torch.randn(50, 8, generator=g, requires_grad=True)
    index = torch.randint(0, 7, (50,), generator=g)
    ops.scatter_sum(src, index, dim=0, dim_size=7).sum().backward()
    assert torch.equal(src.grad, torch.ones_like(src))
    out, arg = ops.scatter_max(src.detach(), index, 0, None, 7)
    ref = torch.full((7, 8), float('-inf')).scatter_reduce(0, index[:, None].expand(50, 8), src.detach(), 'amax')
    assert torch.equal(out, torch.where(ref == float('-inf'), torch.zeros(()), ref))
    mean = ops.scatter_mean(src.detach(), index, 0, None, 7)
    cnt = torch.bincount(index, minlength=7).clamp(min=1)[:, None]
    torch.testing.assert_close(mean, torch.zeros(7, 8).index_add_(0, index, src.detach()) / cnt)
    indptr = torch.tensor([0, 10, 10, 35, 50])
    s = src.detach().clone().requires_grad_()
    ops.segment_mean_csr(s, indptr).sum().backward()
    lens = (indptr[1:] - indptr[:-1]).clamp(min=1).float()
    torch.testing.assert_close(s.grad, (1 / lens).repeat_interleave(indptr[1:] - indptr[:-1])[:, None].expand(50, 8))
    x = torch.randn(50, 3, generator=g, requires_grad=True)
    y = ops.softmax_csr(x, indptr, 0)
    torch.testing.assert_close(y[10:35].sum(0), torch.ones(3))
    y[10:35, 0].sum().backward()
    assert torch.isfinite(x.grad).all()
    with pytest.raises(RuntimeError):
        ops.scatter_sum(src.detach(), torch.tensor([9] * 50), 0, None, 7)   # out-of-range index: checked on CPU
