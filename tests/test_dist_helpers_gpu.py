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


ET = ('paper', 'to', 'paper')
REL = 'paper__to__paper'


@pytest.mark.parametrize('case', G.HETERO_RELABEL_CASES, ids=[c['name'] for c in G.HETERO_RELABEL_CASES])
def test_hetero_relabel_golden(case):
    kw = case['kwargs']
    batch = {'paper': dev(case['batch'])} if 'batch' in case else None
    row, col = torch.ops.pyg.hetero_relabel_neighborhood(
        ['paper'], [ET], {'paper': dev(case['seed'])}, {'paper': dev(case['sampled'])}, {REL: case['counts']},
        {'paper': 6}, batch, kw.get('csc', False), kw.get('disjoint', False))
    assert row[REL].tolist() == case['row'] and col[REL].tolist() == case['col']


def _sample_to_relabel_inputs(node_types, edge_types, out, csc):
    """Turns a hetero_neighbor_sample result (oracle form) into hetero_relabel_neighborhood's inputs: per node
    type the sampled global ids in processing order (layer, edge type, source), per edge type and layer the
    number of neighbours of every source node of that layer's range."""
    rows, cols, nodes, _, nhops, ehops = out[:6]
    L = len(next(iter(ehops.values())))
    sampled = {t: [] for t in node_types}
    batches = {t: [] for t in node_types}
    counts = {e: [] for e in edge_types}
    start = {e: 0 for e in edge_types}
    lo = {t: 0 for t in node_types}
    for ell in range(L):
        hi = {t: lo[t] + nhops[t][ell] for t in node_types}
        for e in edge_types:
            src, dst = (e[0], e[2]) if not csc else (e[2], e[0])
            r, c = (rows[e], cols[e]) if not csc else (cols[e], rows[e])
            n = ehops[e][ell]
            r_l, c_l = r[start[e]:start[e] + n], c[start[e]:start[e] + n]
            start[e] += n
            counts[e].append(np.bincount(r_l - lo[src], minlength=hi[src] - lo[src]).tolist() if hi[src] > lo[src] else [])
            nd = nodes[dst]
            if nd.ndim == 2:
                batches[dst].extend(nd[c_l, 0].tolist())
                sampled[dst].extend(nd[c_l, 1].tolist())
            else:
                sampled[dst].extend(nd[c_l].tolist())
        lo = hi
    return sampled, batches, counts


@pytest.mark.parametrize('csc', [False, True])
@pytest.mark.parametrize('disjoint', [False, True])
def test_hetero_relabel_reproduces_hetero_sample(csc, disjoint):
    # the reference's own consistency check (test_dist_relabel.cpp:123-137) on a 2-type / 3-relation graph:
    # relabelling the sampled global ids gives back hetero_neighbor_sample's (row, col)
    rng = np.random.default_rng(11)
    sizes = {'a': 600, 'b': 400}
    node_types = ['a', 'b']
    ets = [('a', 'x', 'b'), ('b', 'y', 'a'), ('a', 'z', 'a')]
    rowptr, col = {}, {}
    for e in ets:
        n_row, n_col = (sizes[e[0]], sizes[e[2]]) if not csc else (sizes[e[2]], sizes[e[0]])
        deg = rng.poisson(5, n_row).astype(np.int64)
        rowptr[e] = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        col[e] = rng.integers(0, n_col, int(rowptr[e][-1]), dtype=np.int64)
    seeds = {'a': rng.choice(600, 40, replace=False).astype(np.int64), 'b': rng.choice(400, 25, replace=False).astype(np.int64)}
    fan = {ets[0]: [3, 2, 2], ets[1]: [2, 3, 1], ets[2]: [4, 1, 2]}
    out = oracle.hetero_neighbor_sample(node_types, ets, rowptr, col, seeds, fan, csc=csc, disjoint=disjoint, rng_seed=21)
    sampled, batches, counts = _sample_to_relabel_inputs(node_types, ets, out, csc)
    ref_row, ref_col = oracle.hetero_relabel_neighborhood(
        node_types, ets, seeds, sampled, counts, batch_dict=batches if disjoint else None, csc=csc, disjoint=disjoint)
    rel = lambda e: '__'.join(e)
    row, colo = torch.ops.pyg.hetero_relabel_neighborhood(
        node_types, ets, {t: dev(s) for t, s in seeds.items()}, {t: dev(v) for t, v in sampled.items()},
        {rel(e): counts[e] for e in ets}, {t: sizes[t] for t in node_types},
        {t: dev(v) for t, v in batches.items()} if disjoint else None, csc, disjoint)
    for e in ets:
        assert row[rel(e)].cpu().tolist() == ref_row[e].tolist()
        assert colo[rel(e)].cpu().tolist() == ref_col[e].tolist()
        # ... and both equal the sampler's own relabelling
        assert ref_row[e].tolist() == out[0][e].tolist() and ref_col[e].tolist() == out[1][e].tolist()
    assert sum(len(v) for v in out[0].values()) > 500
