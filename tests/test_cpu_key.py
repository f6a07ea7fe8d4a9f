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
