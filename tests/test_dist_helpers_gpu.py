"""pyg::relabel_neighborhood and pyg::merge_sampler_outputs on the HIP device: the golden vectors of the
reference's test/csrc/sampler/test_dist_relabel.cpp / test_dist_merge_outputs.cpp, the reference's own
consistency check (relabelling dist_neighbor_sample's output reproduces neighbor_sample), and random inputs
against the oracle restatement."""
import numpy as np
import pytest
import torch

import oracle
import pyg_lib_amd  # noqa: F401
from pyg_lib_amd import sampler
from tests.golden import sampler_reference_vectors as G

pytestmark = pytest.mark.gpu
dev = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int64)).cuda()


@pytest.mark.parametrize('case', G.RELABEL_CASES, ids=[c['name'] for c in G.RELABEL_CASES])
def test_relabel_golden(case):
    batch = dev(case['batch']) if case['batch'] is not None else None
    row, col = torch.ops.pyg.relabel_neighborhood(dev(case['seed']), dev(case['sampled']), case['counts'], case['num_nodes'],
                                                  batch, False, case['disjoint'])
    assert row.tolist() == case['row'] and col.tolist() == case['col']
    col2, row2 = torch.ops.pyg.relabel_neighborhood(dev(case['seed']), dev(case['sampled']), case['counts'],
                                                    case['num_nodes'], batch, True, case['disjoint'])
    assert row2.tolist() == case['row'] and col2.tolist() == case['col']  # csc only swaps the outputs


@pytest.mark.parametrize('case', G.MERGE_CASES, ids=[c['name'] for c in G.MERGE_CASES])
def test_merge_golden(case):
    batch = dev(case['batch']) if case['batch'] is not None else None
    n, e, b, cnt = torch.ops.pyg.merge_sampler_outputs([dev(x) for x in case['node_ids']], [dev(x) for x in case['edge_ids']],
                                                       case['cumsum'], case['partition_ids'], case['partition_orders'],
                                                       case['num_partitions'], case['num_neighbors'], batch, case['disjoint'])
    assert n.tolist() == case['nodes'] and e.tolist() == case['edges'] and list(cnt) == case['counts']
    assert (b is None) == (case['out_batch'] is None) and (b is None or b.tolist() == case['out_batch'])


@pytest.mark.parametrize('disjoint', [False, True])
def test_relabel_of_dist_sample_reproduces_neighbor_sample(disjoint):
    # test_dist_relabel.cpp:28-36, 64-79 on a random graph: one hop of dist_neighbor_sample + relabel_neighborhood
    # == one hop of neighbor_sample
    rng = np.random.default_rng(5)
    n = 4000
    deg = rng.poisson(9, n).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    seeds = rng.permutation(n)[:300].astype(np.int64)
    rp, cl, sd = dev(rowptr), dev(col), dev(seeds)
    torch.manual_seed(3)
    ref = sampler.neighbor_sample(rp, cl, sd, [6], disjoint=disjoint)
    torch.manual_seed(3)
    node, edge, cumsum = torch.ops.pyg.dist_neighbor_sample(rp, cl, sd, 6, None, None, None, None, True, False, True, disjoint,
                                                            'uniform')
    S = seeds.size
    if disjoint:
        batch, sampled = node[S:, 0].contiguous(), node[S:, 1].contiguous()
    else:
        batch, sampled = None, node[S:].contiguous()
    counts = [cumsum[i + 1] - cumsum[i] for i in range(S)]
    row, colo = torch.ops.pyg.relabel_neighborhood(sd, sampled, counts, n, batch, False, disjoint)
    assert torch.equal(row, ref[0]) and torch.equal(colo, ref[1])


def test_random_against_oracle():
    rng = np.random.default_rng(6)
    for disjoint in (False, True):
        S, nsrc = 500, 500
        seed = rng.integers(0, 3000, S)
        counts = rng.integers(0, 12, nsrc).tolist()
        E = int(sum(counts))
        sampled = rng.integers(0, 3000, E)
        batch = np.repeat(np.arange(nsrc), counts) if disjoint else None
        want = oracle.relabel_neighborhood(seed, sampled, counts, 3000, batch, False, disjoint)
        got = torch.ops.pyg.relabel_neighborhood(dev(seed), dev(sampled), counts, 3000, dev(batch) if disjoint else None, False,
                                                 disjoint)
        assert np.array_equal(got[0].cpu().numpy(), want[0]) and np.array_equal(got[1].cpu().numpy(), want[1])
    # merge: 4 partitions, random segment lengths
    P = 4
    cums, nodes, edges = [], [], []
    for p in range(P):
        lens = rng.integers(0, 9, 200)
        first = int(rng.integers(1, 50))
        cs = np.concatenate([[first], first + np.cumsum(lens)]).tolist()
        cums.append(cs)
        nodes.append(rng.integers(0, 10 ** 6, cs[-1]))
        edges.append(rng.integers(0, 10 ** 6, cs[-1] - first))
    pids, pords = [], []
    for p in range(P):
        for o in range(200):
            pids.append(p)
            pords.append(o)
    perm = rng.permutation(len(pids))
    pids, pords = [pids[i] for i in perm], [pords[i] for i in perm]
    batch = rng.integers(0, 77, len(pids))
    want = oracle.merge_sampler_outputs(nodes, edges, cums, pids, pords, P, 8, batch, True)
    got = torch.ops.pyg.merge_sampler_outputs([dev(x) for x in nodes], [dev(x) for x in edges], cums, pids, pords, P, 8, dev(batch),
                                              True)
    assert np.array_equal(got[0].cpu().numpy(), want[0]) and np.array_equal(got[1].cpu().numpy(), want[1])
    assert np.array_equal(got[2].cpu().numpy(), want[2]) and list(got[3]) == want[3]
