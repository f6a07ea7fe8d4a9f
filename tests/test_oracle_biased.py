"""Biased (edge_weight) sampling: the oracle against vectors computed with the real libtorch ops
(tests/golden/make_biased_golden.py) and against the reference's own biased tests
(test/csrc/sampler/test_neighbor.cpp:300-377, weights {1, 0} on the cycle graph)."""
import numpy as np
import pytest

import oracle
from tests.golden import biased_cases
from tests.golden.sampler_reference_vectors import cycle_graph

CASES = biased_cases.load()


@pytest.mark.parametrize('case', CASES, ids=[f"c{c['id']}" for c in CASES])
def test_oracle_matches_torch_vectors(case):
    out = oracle.hetero_neighbor_sample(case['node_types'], case['edge_types'], case['rowptr'], case['col'],
                                        case['seed'], case['fan'], disjoint=case['disjoint'], replace=case['replace'],
                                        rng_seed=case['manual_seed'], edge_weight_dict=case['weight'])
    rows, cols, nodes, eids, nh, eh, info = out
    for e in case['edge_types']:
        np.testing.assert_array_equal(rows[e], case['row_out'][e])
        np.testing.assert_array_equal(cols[e], case['col_out'][e])
        np.testing.assert_array_equal(eids[e], case['edge_out'][e])
        assert eh[e] == case['ehops'][e]
    for t in case['node_types']:
        np.testing.assert_array_equal(nodes[t], case['node'][t].reshape(nodes[t].shape))
        assert nh[t] == case['nhops'][t]
    assert info['rng_blocks'] == 1  # biased rows never touch the prefetched engine


def test_reference_biased_test_vectors():
    # BiasedNeighborTest :300-328 -- only the weight-1 (even) edges can be drawn, whatever the random numbers
    rowptr, col = cycle_graph(6)
    w = np.tile(np.array([1.0, 0.0], dtype=np.float32), 6)
    for seed in (0, 1, 123456):
        row, c, node, eid, _, _, _ = oracle.neighbor_sample(rowptr, col, np.array([0, 1]), [1], edge_weight=w, rng_seed=seed)
        assert row.tolist() == [0, 1] and c.tolist() == [2, 0]
        assert node.tolist() == [0, 1, 5] and eid.tolist() == [0, 2]


def test_reference_hetero_biased_test_vectors():
    # HeteroBiasedNeighborTest :330-377
    rowptr, col = cycle_graph(6)
    w = np.tile(np.array([1.0, 0.0], dtype=np.float32), 6)
    et = ('paper', 'to', 'paper')
    out = oracle.hetero_neighbor_sample(['paper'], [et], {et: rowptr}, {et: col}, {'paper': np.array([0, 1])},
                                        {et: [1]}, edge_weight_dict={et: w}, rng_seed=7)
    assert out[0][et].tolist() == [0, 1] and out[1][et].tolist() == [2, 0]
    assert out[2]['paper'].tolist() == [0, 1, 5] and out[3][et].tolist() == [0, 2]


def test_topk_tie_order_is_libstdcxx():
    # all keys equal: the order is whatever partial_sort / nth_element + sort leave (not index order)
    k = np.zeros(200, dtype=np.float32)
    a = oracle.topk_desc(k, 3)       # 3 * 64 <= 200: partial_sort
    b = oracle.topk_desc(k, 50)      # nth_element + sort
    import torch
    assert a.tolist() == torch.zeros(200).topk(3)[1].tolist()
    assert b.tolist() == torch.zeros(200).topk(50)[1].tolist()
    rng = np.random.default_rng(0)
    for n, kk in ((5, 2), (17, 16), (64, 1), (300, 4), (300, 5), (1000, 999), (4096, 64), (4097, 65)):
        x = rng.integers(0, 4, n).astype(np.float32)
        x[rng.random(n) < 0.1] = -np.inf
        assert oracle.topk_desc(x, kk).tolist() == torch.from_numpy(x).topk(kk)[1].tolist()
        xd = x.astype(np.float64)
        assert oracle.topk_desc(xd, kk).tolist() == torch.from_numpy(xd).topk(kk)[1].tolist()


def test_mixed_weighted_and_uniform_relations_share_one_generator():
    # a weighted relation draws from the generator between the engine's prefetches (neighbor_kernel.cpp:732-760)
    rng = np.random.default_rng(3)
    n = 80
    deg = rng.integers(3, 12, n)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, rowptr[-1]).astype(np.int64)
    e1, e2 = ('n', 'w', 'n'), ('n', 'u', 'n')
    w = rng.random(rowptr[-1]).astype(np.float32) + 0.1
    out = oracle.hetero_neighbor_sample(['n'], [e1, e2], {e1: rowptr, e2: rowptr}, {e1: col, e2: col},
                                        {'n': np.arange(6)}, {e1: [2, 2], e2: [2, 2]}, rng_seed=11,
                                        edge_weight_dict={e1: w})
    info = out[6]
    assert info['rng_raw_draws'] > 0 and info['rng_draws'] > 0
    assert all(len(out[0][e]) > 0 for e in (e1, e2))
