"""Biased (edge_weight) neighbour sampling on the device: exact parity with the vectors computed with the real
libtorch ops (tests/golden/biased_golden.npz), with the reference's own biased tests
(test/csrc/sampler/test_neighbor.cpp:300-377) and with the oracle on larger graphs, including rows whose keys
tie (zero weights, equal weights) -- those follow libstdc++'s partial_sort / nth_element order."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from pyg_lib_amd import _capi, sampler
from tests.golden import biased_cases
from tests.golden.sampler_reference_vectors import cycle_graph

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
I64_MIN, I64_MAX = -2**63, 2**63 - 1
CASES = biased_cases.load()


def dev(a, dtype=torch.long):
    return torch.as_tensor(np.asarray(a)).to(dtype).to(DEV)


def wdev(w):
    return torch.from_numpy(np.ascontiguousarray(w)).to(DEV)


def rel(e):
    return '__'.join(e)


@pytest.mark.parametrize('case', CASES, ids=[f"c{c['id']}" for c in CASES])
def test_torch_vectors(case):
    torch.manual_seed(case['manual_seed'])
    ets = case['edge_types']
    if not case['hetero']:
        e = ets[0]
        row, col, node, eid, nh, eh = sampler.neighbor_sample(
            dev(case['rowptr'][e]), dev(case['col'][e]), dev(case['seed']['n']), case['fan'][e],
            edge_weight=wdev(case['weight'][e]), disjoint=case['disjoint'], replace=case['replace'])
        rows, cols, nodes, eids, nhs, ehs = {e: row}, {e: col}, {'n': node}, {e: eid}, {'n': nh}, {e: eh}
    else:
        rows, cols, nodes, eids, nhs, ehs = sampler.hetero_neighbor_sample(
            {e: dev(case['rowptr'][e]) for e in ets}, {e: dev(case['col'][e]) for e in ets},
            {t: dev(s) for t, s in case['seed'].items()}, case['fan'],
            edge_weight_dict={e: wdev(case['weight'][e]) for e in ets}, disjoint=case['disjoint'],
            replace=case['replace'])
    for e in ets:
        assert rows[e].cpu().tolist() == case['row_out'][e].tolist()
        assert cols[e].cpu().tolist() == case['col_out'][e].tolist()
        assert eids[e].cpu().tolist() == case['edge_out'][e].tolist()
        assert list(ehs[e]) == case['ehops'][e]
    for t in case['node_types']:
        assert nodes[t].cpu().numpy().reshape(-1).tolist() == case['node'][t].reshape(-1).tolist()
        assert list(nhs[t]) == case['nhops'][t]


def test_reference_biased_tests():
    rowptr, col = cycle_graph(6)
    w = np.tile(np.array([1.0, 0.0], dtype=np.float32), 6)
    # BiasedNeighborTest :300-328
    row, c, node, eid, _, _ = sampler.neighbor_sample(dev(rowptr), dev(col), dev([0, 1]), [1], edge_weight=wdev(w))
    assert row.cpu().tolist() == [0, 1] and c.cpu().tolist() == [2, 0]
    assert node.cpu().tolist() == [0, 1, 5] and eid.cpu().tolist() == [0, 2]
    # HeteroBiasedNeighborTest :330-377
    et = ('paper', 'to', 'paper')
    out = sampler.hetero_neighbor_sample({et: dev(rowptr)}, {et: dev(col)}, {'paper': dev([0, 1])}, {et: [1]},
                                         edge_weight_dict={et: wdev(w)})
    assert out[0][et].cpu().tolist() == [0, 1] and out[1][et].cpu().tolist() == [2, 0]
    assert out[2]['paper'].cpu().tolist() == [0, 1, 5] and out[3][et].cpu().tolist() == [0, 2]


def big_graph(seed, n=20000, avg=30, hubs=(3000, 9000, 70000)):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg, n).astype(np.int64)
    deg[rng.random(n) < 0.05] = 0
    for h in hubs:
        deg[rng.integers(0, n)] = h
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    return rowptr, col, rng


@pytest.mark.parametrize('kind', ['random', 'ones', 'zeros_mixed', 'small_ints'])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_against_oracle_large(kind, dtype):
    rowptr, col, rng = big_graph(hash(kind) % 1000)
    E = col.size
    if kind == 'random':
        w = rng.random(E) + 0.01
    elif kind == 'ones':
        w = np.ones(E)
    elif kind == 'zeros_mixed':
        w = (rng.random(E) < 0.3).astype(np.float64)
    else:
        w = rng.integers(1, 4, E).astype(np.float64)
    w = w.astype(dtype)
    seeds = rng.choice(rowptr.size - 1, 256, replace=False).astype(np.int64)
    # make sure the hubs are expanded in the first hop
    seeds[:3] = np.argsort(np.diff(rowptr))[-3:]
    ms = 4242
    for fan, disjoint in (([10, 5], False), ([40, 3], True), ([100], False)):
        torch.manual_seed(ms)
        out = sampler.neighbor_sample(dev(rowptr), dev(col), dev(seeds), fan, edge_weight=wdev(w), disjoint=disjoint)
        after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
        ref = oracle.neighbor_sample(rowptr, col, seeds, fan, edge_weight=w, disjoint=disjoint, rng_seed=ms)
        row, c, node, eid, nh, eh = out
        rrow, rcol, rnode, reid, rnh, reh, info = ref
        assert nh == rnh and eh == reh
        assert torch.equal(eid.cpu(), torch.from_numpy(reid))
        assert torch.equal(row.cpu(), torch.from_numpy(rrow))
        assert torch.equal(c.cpu(), torch.from_numpy(rcol))
        assert torch.equal(node.cpu(), torch.from_numpy(rnode))
        # the global CPU generator advanced by the engine's prefetch + every uniform_ draw
        assert info['rng_blocks'] == 1 and info['rng_raw_draws'] > 0
        assert after == oracle.mt19937_word_after(ms, 256 + info['rng_raw_draws'])


@pytest.mark.parametrize('csc', [False, True])
def test_hetero_against_oracle(csc):
    rng = np.random.default_rng(5)
    sizes = {'a': 5000, 'b': 3000}
    ets = [('a', 'x', 'b'), ('b', 'y', 'a'), ('a', 'z', 'a')]
    rowptr, col, w = {}, {}, {}
    for i, e in enumerate(ets):
        n_row, n_col = (sizes[e[0]], sizes[e[2]]) if not csc else (sizes[e[2]], sizes[e[0]])  # csc: (colptr, row)
        deg = rng.poisson(20, n_row).astype(np.int64)
        deg[rng.integers(0, deg.size, 3)] = 2500
        rowptr[e] = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        col[e] = rng.integers(0, n_col, int(rowptr[e][-1]), dtype=np.int64)
        w[e] = (rng.integers(0, 3, col[e].size)).astype(np.float32 if i != 1 else np.float64)
    seeds = {'a': rng.choice(5000, 64, replace=False).astype(np.int64), 'b': rng.choice(3000, 32, replace=False).astype(np.int64)}
    fan = {ets[0]: [5, 3, 2], ets[1]: [4, 4, 1], ets[2]: [8, 0, 3]}
    torch.manual_seed(99)
    out = sampler.hetero_neighbor_sample({e: dev(rowptr[e]) for e in ets}, {e: dev(col[e]) for e in ets},
                                         {t: dev(s) for t, s in seeds.items()}, fan,
                                         edge_weight_dict={e: wdev(w[e]) for e in ets}, csc=csc)
    after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
    node_types = ['a', 'b']
    ref = oracle.hetero_neighbor_sample(node_types, ets, rowptr, col, seeds, fan, csc=csc, rng_seed=99,
                                        edge_weight_dict=w)
    assert sum(len(v) for v in ref[0].values()) > 1000
    for e in ets:
        assert torch.equal(out[0][e].cpu(), torch.from_numpy(ref[0][e]))
        assert torch.equal(out[1][e].cpu(), torch.from_numpy(ref[1][e]))
        assert torch.equal(out[3][e].cpu(), torch.from_numpy(ref[3][e]))
        assert list(out[5][e]) == ref[5][e]
    for t in node_types:
        assert torch.equal(out[2][t].cpu(), torch.from_numpy(ref[2][t]))
    assert after == oracle.mt19937_word_after(99, 256 + ref[6]['rng_raw_draws'])


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_tie_rows_follow_libstdcxx_order(dtype):
    """Rows whose keys tie (all / most weights zero -> -inf keys) in both of topk's branches: partial_sort
    (count * 64 <= degree) and nth_element + sort, across the introselect / heap / insertion-sort regimes."""
    rng = np.random.default_rng(17)
    degs = [2, 3, 4, 7, 16, 17, 33, 64, 65, 100, 128, 129, 640, 641, 1000, 4096, 5000, 20000]
    n = len(degs)
    rowptr = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    seeds = np.arange(n, dtype=np.int64)
    for kind in ('all_zero', 'three_positive', 'two_values'):
        if kind == 'all_zero':
            w = np.zeros(col.size)
        elif kind == 'three_positive':
            w = np.zeros(col.size)
            for r in range(n):
                w[rowptr[r] + rng.choice(degs[r], min(3, degs[r]), replace=False)] = 1.0
        else:
            w = np.where(rng.random(col.size) < 0.5, 0.0, np.inf)  # keys: -inf or -0.0
        w = w.astype(dtype)
        for k in (1, 2, 5, 17, 64, 100, 313):
            torch.manual_seed(k)
            out = sampler.neighbor_sample(dev(rowptr), dev(col), dev(seeds), [k], edge_weight=wdev(w))
            ref = oracle.neighbor_sample(rowptr, col, seeds, [k], edge_weight=w, rng_seed=k)
            assert torch.equal(out[3].cpu(), torch.from_numpy(ref[3])), (kind, k)
            assert torch.equal(out[2].cpu(), torch.from_numpy(ref[2])), (kind, k)
            assert torch.equal(out[1].cpu(), torch.from_numpy(ref[1])), (kind, k)


@pytest.mark.parametrize('disjoint', [False, True])
def test_mixed_weighted_and_uniform_relations(disjoint):
    """Only some relations weighted (neighbor_kernel.cpp:732-760): the weighted ones draw straight from the
    generator BETWEEN the engine's 128-word prefetches, so later prefetched blocks lie further on in the stream."""
    rng = np.random.default_rng(23)
    sizes = {'a': 3000, 'b': 2000}
    ets = [('a', 'u1', 'b'), ('b', 'w1', 'a'), ('a', 'w2', 'a'), ('b', 'u2', 'b')]
    rowptr, col, w = {}, {}, {}
    for e in ets:
        deg = rng.poisson(14, sizes[e[0]]).astype(np.int64)
        deg[rng.integers(0, deg.size, 2)] = 1500
        rowptr[e] = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        col[e] = rng.integers(0, sizes[e[2]], int(rowptr[e][-1]), dtype=np.int64)
    w[ets[1]] = (rng.random(col[ets[1]].size) + 0.1).astype(np.float32)
    w[ets[2]] = rng.integers(0, 3, col[ets[2]].size).astype(np.float64)
    seeds = {'a': rng.choice(3000, 150, replace=False).astype(np.int64), 'b': rng.choice(2000, 90, replace=False).astype(np.int64)}
    refills = []
    for fan in ({ets[0]: [6, 4, 3], ets[1]: [5, 3, 2], ets[2]: [3, 5, 2], ets[3]: [4, 2, 4]},
                {ets[0]: [70, -1], ets[1]: [2, 2], ets[2]: [1, 80], ets[3]: [3, 0]}):
        torch.manual_seed(31)
        out = sampler.hetero_neighbor_sample({e: dev(rowptr[e]) for e in ets}, {e: dev(col[e]) for e in ets},
                                             {t: dev(s) for t, s in seeds.items()}, fan,
                                             edge_weight_dict={e: wdev(v) for e, v in w.items()}, disjoint=disjoint)
        after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
        ref = oracle.hetero_neighbor_sample(['a', 'b'], ets, rowptr, col, seeds, fan, rng_seed=31, edge_weight_dict=w,
                                            disjoint=disjoint)
        info = ref[6]
        assert info['rng_raw_draws'] > 0
        refills.append(info['rng_blocks'])
        for e in ets:
            assert torch.equal(out[3][e].cpu(), torch.from_numpy(ref[3][e])), e
            assert torch.equal(out[0][e].cpu(), torch.from_numpy(ref[0][e]))
            assert torch.equal(out[1][e].cpu(), torch.from_numpy(ref[1][e]))
        for t in ('a', 'b'):
            assert torch.equal(out[2][t].cpu(), torch.from_numpy(ref[2][t]))
        assert after == oracle.mt19937_word_after(31, 256 * info['rng_blocks'] + info['rng_raw_draws'])
    assert max(refills) > 10  # the engine refilled many times between the weighted relations' draws


def test_log_f32_matches_on_every_uniform_argument():
    lib = _capi.lib()
    lib.pyg_hip_biased_log_f32.restype = ctypes.c_int
    lib.pyg_hip_biased_log_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    u = (torch.arange(0, 1 << 24, dtype=torch.int64).double() * 2.0**-24).float()
    ud = u.to(DEV)
    out = torch.empty_like(ud)
    rc = lib.pyg_hip_biased_log_f32(ud.data_ptr(), out.data_ptr(), ud.numel(), None)
    assert rc == 0
    torch.cuda.synchronize()
    ref = torch.from_numpy(oracle.biased_log_f32(u.numpy()))
    got = out.cpu()
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32))


def test_unsupported_modes_fail_loudly():
    rowptr, col = cycle_graph(6)
    w = wdev(np.ones(12, dtype=np.float32))
    with pytest.raises(RuntimeError, match='float32 or float64'):
        sampler.neighbor_sample(dev(rowptr), dev(col), dev([0, 1]), [1], edge_weight=w.half())


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('disjoint', [False, True])
def test_biased_dist_neighbor_sample(dtype, disjoint):
    # biased_sample in distributed mode (neighbor_kernel.cpp:436-447 with :296-303): one hop, no relabelling
    rowptr, col, rng = big_graph(9, n=6000, avg=20, hubs=(900, 5000))
    w = rng.integers(0, 3, col.size).astype(dtype) if disjoint else (rng.random(col.size) + 0.05).astype(dtype)
    seeds = rng.choice(6000, 400, replace=False).astype(np.int64)
    seeds[:2] = np.argsort(np.diff(rowptr))[-2:]
    for fan in (5, 64, -1):
        torch.manual_seed(8)
        node, edge, cumsum = torch.ops.pyg.dist_neighbor_sample(dev(rowptr), dev(col), dev(seeds), fan, None, None, None,
                                                                wdev(w), True, False, True, disjoint, 'uniform')
        after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
        rnode, redge, rcumsum, info = oracle.dist_neighbor_sample(rowptr, col, seeds, fan, rng_seed=8, edge_weight=w,
                                                                  disjoint=disjoint)
        assert torch.equal(edge.cpu(), torch.from_numpy(redge))
        assert torch.equal(node.cpu(), torch.from_numpy(rnode))
        assert cumsum == rcumsum
        assert after == oracle.mt19937_word_after(8, 256 * info['rng_blocks'] + info['rng_raw_draws'])


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_biased_with_replacement_against_oracle(dtype):
    """replace=True: at::multinomial(weight, count, true) for count > 1 (sequential cumulative sum in the weights'
    type, one double per sample); a fan-out of 1 and invalid distributions fail loudly."""
    rowptr, col, rng = big_graph(31, n=8000, avg=25, hubs=(2000, 30000))
    w = (rng.random(col.size) ** 3 + 1e-3).astype(dtype)
    w[rng.random(col.size) < 0.2] = 0  # zero weights are fine as long as a row keeps a positive one
    deg = np.diff(rowptr)
    for v in range(deg.size):  # ... so give every row one
        if deg[v]:
            w[rowptr[v]] = max(w[rowptr[v]], dtype(0.5))
    seeds = rng.choice(8000, 300, replace=False).astype(np.int64)
    seeds[:2] = np.argsort(deg)[-2:]
    for fan, disjoint in (([6, 3, 2], False), ([40, 2], True), ([-1, 5], False), ([200], False)):
        torch.manual_seed(77)
        out = sampler.neighbor_sample(dev(rowptr), dev(col), dev(seeds), fan, edge_weight=wdev(w), replace=True,
                                      disjoint=disjoint)
        after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
        ref = oracle.neighbor_sample(rowptr, col, seeds, fan, edge_weight=w, replace=True, disjoint=disjoint, rng_seed=77)
        for k in range(4):
            assert torch.equal(out[k].cpu(), torch.from_numpy(ref[k])), (fan, k)
        assert out[4] == ref[4] and out[5] == ref[5]
        assert after == oracle.mt19937_word_after(77, 256 * ref[6]['rng_blocks'] + ref[6]['rng_raw_draws'])
    # a fan-out of 1: at::multinomial's single-draw route (exponential_ + argmax)
    for fan in ([1], [3, 1, 1], [1, 4]):
        torch.manual_seed(78)
        out = sampler.neighbor_sample(dev(rowptr), dev(col), dev(seeds), fan, edge_weight=wdev(w), replace=True)
        after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
        ref = oracle.neighbor_sample(rowptr, col, seeds, fan, edge_weight=w, replace=True, rng_seed=78)
        for k in range(4):
            assert torch.equal(out[k].cpu(), torch.from_numpy(ref[k])), (fan, k)
        assert after == oracle.mt19937_word_after(78, 256 * ref[6]['rng_blocks'] + ref[6]['rng_raw_draws'])
    bad = w.copy()
    bad[rowptr[seeds[5]]:rowptr[seeds[5] + 1]] = 0  # one sampled row without any positive weight
    with pytest.raises(RuntimeError, match='invalid multinomial distribution'):
        sampler.neighbor_sample(dev(rowptr), dev(col), dev(seeds), [3], edge_weight=wdev(bad), replace=True)
    with pytest.raises(RuntimeError, match='invalid multinomial distribution'):
        oracle.neighbor_sample(rowptr, col, seeds, [3], edge_weight=bad, replace=True)


def test_biased_dist_neighbor_sample_with_replacement():
    rowptr, col, rng = big_graph(4, n=5000, avg=15, hubs=(3000,))
    w = (rng.random(col.size) + 0.05).astype(np.float32)
    seeds = rng.choice(5000, 300, replace=False).astype(np.int64)
    for fan, disjoint in ((4, False), (70, True)):
        torch.manual_seed(9)
        node, edge, cumsum = torch.ops.pyg.dist_neighbor_sample(dev(rowptr), dev(col), dev(seeds), fan, None, None, None,
                                                                wdev(w), True, True, True, disjoint, 'uniform')
        after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
        rnode, redge, rcumsum, info = oracle.dist_neighbor_sample(rowptr, col, seeds, fan, rng_seed=9, edge_weight=w,
                                                                  replace=True, disjoint=disjoint)
        assert torch.equal(edge.cpu(), torch.from_numpy(redge)) and torch.equal(node.cpu(), torch.from_numpy(rnode))
        assert cumsum == rcumsum
        assert after == oracle.mt19937_word_after(9, 256 * info['rng_blocks'] + info['rng_raw_draws'])
