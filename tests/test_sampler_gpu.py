"""Bit-exact parity of the HIP neighbour sampler with the oracle and the reference's golden vectors.

Modelled on test/csrc/sampler/test_neighbor.cpp; every comparison is exact (integer outputs,
SURVEY.md 8(a) S1-S5), including how far torch's global CPU generator advanced.
"""
import numpy as np
import pytest
import torch

import oracle
from pyg_lib_amd import sampler
from tests.golden import sampler_reference_vectors as G

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
I64_MIN, I64_MAX = -2**63, 2**63 - 1


def dev(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.long).to(DEV)


def random_csr(n, avg_deg, seed, max_deg=None, zero_frac=0.1):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg_deg, n).astype(np.int64)
    deg[rng.random(n) < zero_frac] = 0
    if max_deg is not None:
        deg = np.minimum(deg, max_deg)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    return rowptr, col


def run_both(rowptr, col, seed, fanout, manual_seed, **kw):
    torch.manual_seed(manual_seed)
    dkw = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    out = sampler.neighbor_sample(dev(rowptr), dev(col), dev(seed), fanout, **dkw)
    after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
    ref = oracle.neighbor_sample(rowptr, col, np.asarray(seed, dtype=np.int64), fanout, rng_seed=manual_seed, **kw)
    return out, after, ref


def assert_same(out, after, ref, manual_seed, return_edge_id=True):
    row, col, node, eid, nh, eh = out
    rrow, rcol, rnode, reid, rnh, reh, info = ref
    assert nh == rnh and eh == reh
    assert torch.equal(row.cpu(), torch.from_numpy(rrow))
    assert torch.equal(col.cpu(), torch.from_numpy(rcol))
    assert torch.equal(node.cpu(), torch.from_numpy(rnode))
    if return_edge_id:
        assert torch.equal(eid.cpu(), torch.from_numpy(reid))
    else:
        assert eid is None
    # the global CPU generator advanced by exactly the reference's number of 128-word prefetches
    expect_after = int(oracle.mt19937_words(manual_seed, info['rng_blocks'] * 128 + 1)[-1])
    assert after == expect_after


SUPPORTED = list(G.CASES)  # all ten non-biased golden tests of test_neighbor.cpp


@pytest.mark.parametrize('case', SUPPORTED, ids=[c['name'] for c in SUPPORTED])
def test_reference_golden_vectors(case):
    torch.manual_seed(case.get('manual_seed', 0))
    kw = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in case['kwargs'].items()}
    row, col, node, eid, nh, eh = sampler.neighbor_sample(dev(case['rowptr']), dev(case['col']), dev(case['seed']),
                                                          case['num_neighbors'], **kw)
    assert row.cpu().tolist() == case['row']
    assert col.cpu().tolist() == case['col_out']
    assert node.cpu().tolist() == case['node']
    assert eid.cpu().tolist() == case['edge']
    if 'nodes_per_hop' in case:
        assert nh == case['nodes_per_hop'] and eh == case['edges_per_hop']


def test_reference_hetero_golden_vector():
    c = G.HETERO_CASE
    et = c['edge_types'][0]
    out = sampler.hetero_neighbor_sample({et: dev(G.ROWPTR)}, {et: dev(G.COL)}, {'paper': dev(c['seed'])},
                                         {et: c['num_neighbors']})
    assert out[0][et].cpu().tolist() == c['row']
    assert out[1][et].cpu().tolist() == c['col_out']
    assert out[2]['paper'].cpu().tolist() == c['node']
    assert out[3][et].cpu().tolist() == c['edge']
    assert out[4]['paper'] == c['nodes_per_hop'] and out[5][et] == c['edges_per_hop']


@pytest.mark.parametrize('manual_seed', [0, 12345, 123456])
@pytest.mark.parametrize('variant', ['plain', 'disjoint', 'csc', 'no_edge_id', 'replace', 'replace_disjoint'])
def test_random_graph_matches_oracle(manual_seed, variant):
    rowptr, col = random_csr(5000, 12, seed=1)
    seeds = np.random.default_rng(2).permutation(5000)[:64]
    kw = dict(plain={}, disjoint=dict(disjoint=True), csc=dict(csc=True), no_edge_id=dict(return_edge_id=False),
              replace=dict(replace=True), replace_disjoint=dict(replace=True, disjoint=True))[variant]
    out, after, ref = run_both(rowptr, col, seeds, [15, 10, 5], manual_seed, **kw)
    assert_same(out, after, ref, manual_seed, return_edge_id=kw.get('return_edge_id', True))
    assert sum(ref[5]) > 5000  # a real workload, several RNG refills
    assert ref[6]['rng_blocks'] > 3


def test_duplicate_seeds_and_zero_degree_and_full_neighbourhood():
    rowptr, col = random_csr(300, 6, seed=3, zero_frac=0.3)
    seeds = np.array([5, 5, 7, 5, 9, 7, 0, 1, 2])
    for fan in ([-1, -1], [3, -1, 2], [0, 4], [100]):
        out, after, ref = run_both(rowptr, col, seeds, fan, 7)
        assert_same(out, after, ref, 7)
    out, after, ref = run_both(rowptr, col, np.zeros(0, dtype=np.int64), [3, 3], 7)
    assert_same(out, after, ref, 7)


def test_large_fanout_uses_history_of_earlier_rounds():
    # fan-out > 64: Floyd's chosen set spans several 64-lane rounds
    rowptr, col = random_csr(2000, 400, seed=4, zero_frac=0.0)
    seeds = np.arange(0, 2000, 97)
    out, after, ref = run_both(rowptr, col, seeds, [150, 3], 11)
    assert sampler.last_mode() == 'fused'   # round 5: fan-outs of 65 ... 1024 sample one wave per node INSIDE the fused chain
    assert_same(out, after, ref, 11)
    out, after, ref = run_both(rowptr, col, seeds, [200], 11, replace=True)
    assert sampler.last_mode() == 'fused'
    assert_same(out, after, ref, 11)
    # rows shorter than the fan-out (whole neighbourhoods of up to 1023 edges), a hub, disjoint, three hops
    rng = np.random.default_rng(41)
    n = 3000
    deg = rng.integers(0, 900, n).astype(np.int64)
    deg[11] = 70_000
    rowptr2 = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col2 = rng.integers(0, n, int(rowptr2[-1]), dtype=np.int64)
    for kw in ({}, dict(disjoint=True), dict(replace=True)):
        out, after, ref = run_both(rowptr2, col2, np.array([11, 5, 2999, 40]), [700, 65, 2], 12, **kw)
        assert sampler.last_mode() == 'fused'
        assert_same(out, after, ref, 12)
    out, after, ref = run_both(rowptr2, col2, np.array([11, 5]), [1025, 2], 13)   # beyond the chain's limit: another driver, same bits
    assert sampler.last_mode() != 'fused'
    assert_same(out, after, ref, 13)


def test_wide_draws_above_65535():
    # hubs with degree >= 2^16 switch their draws to 32 bits (rand_engine.h:44-50) in the middle of
    # the stream; neighbours keep 16-bit draws.
    rng = np.random.default_rng(5)
    n = 4000
    deg = rng.poisson(20, n).astype(np.int64)
    deg[17] = 70000
    deg[123] = 65536
    deg[500] = 65535
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    seeds = np.array([17, 3, 123, 500, 9, 17])
    for kw in ({}, dict(replace=True), dict(disjoint=True)):
        out, after, ref = run_both(rowptr, col, seeds, [25, 10], 99, **kw)
        # round 5: the fused chain carries the general transition tables itself -- a hub no longer sends
        # the call to the queued chain; degree 65536 with fan-out 25 mixes 16- and 32-bit draws inside one row
        assert sampler.last_mode() == 'fused'
        assert_same(out, after, ref, 99)


def test_every_row_a_hub_exceeds_the_word_allowance_and_repeats_synchronising():
    """The fused chain speculates four 16-bit draws per random word (+ 64 words per relation and hop for rows of degree
    >= 2^16, whose draws take 32 bits).  A graph in which EVERY row is such a hub needs twice the words: the chain's
    overflow flag sends the call to the synchronising driver -- same bits, same generator state."""
    rng = np.random.default_rng(15)
    n = 500
    deg = np.full(n, 66_000, dtype=np.int64)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    seeds = rng.permutation(n).astype(np.int64)
    out, after, ref = run_both(rowptr, col, seeds, [64, 4], 7)   # 32,000 wide draws = 16,000 words in hop 0: 8,000 + slack speculated
    assert_same(out, after, ref, 7)
    assert sampler.last_mode() == 'synchronising'
    out, after, ref = run_both(rowptr, col, seeds[:200], [12, 6], 7)   # (fewer of them still fit what was generated)
    assert_same(out, after, ref, 7)
    # a few hubs among ordinary rows stay inside the allowance: fused
    deg2 = rng.poisson(12, n).astype(np.int64)
    deg2[[3, 77, 300]] = 66_000
    rowptr2 = np.concatenate([[0], np.cumsum(deg2)]).astype(np.int64)
    col2 = rng.integers(0, n, int(rowptr2[-1]), dtype=np.int64)
    out, after, ref = run_both(rowptr2, col2, np.array([3, 5, 77, 300, 9]), [12, 6], 8)
    assert sampler.last_mode() == 'fused'
    assert_same(out, after, ref, 8)


@pytest.mark.parametrize('csc', [False, True])
@pytest.mark.parametrize('disjoint', [False, True])
def test_hetero_random_graph_matches_oracle(csc, disjoint):
    rng = np.random.default_rng(6)
    sizes = {'a': 700, 'b': 300, 'c': 1100}
    ets = [('a', 'x', 'b'), ('b', 'y', 'a'), ('a', 'z', 'c'), ('c', 'w', 'c'), ('b', 'v', 'c')]
    rp, cl = {}, {}
    for (s, r, d) in ets:
        rows, cols = (sizes[s], sizes[d]) if not csc else (sizes[d], sizes[s])
        deg = rng.poisson(7, rows).astype(np.int64)
        rp[(s, r, d)] = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        cl[(s, r, d)] = rng.integers(0, cols, int(deg.sum()), dtype=np.int64)
    seeds = {'a': rng.permutation(700)[:20].astype(np.int64), 'c': rng.permutation(1100)[:12].astype(np.int64)}
    fan = {e: [8, 4] for e in ets}
    torch.manual_seed(31)
    out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in rp.items()}, {e: dev(v) for e, v in cl.items()},
                                         {k: dev(v) for k, v in seeds.items()}, fan, csc=csc, disjoint=disjoint)
    ref = oracle.hetero_neighbor_sample(['a', 'b', 'c'], ets, rp, cl, seeds, fan, csc=csc, disjoint=disjoint,
                                        rng_seed=31)
    for e in ets:
        assert torch.equal(out[0][e].cpu(), torch.from_numpy(ref[0][e]))
        assert torch.equal(out[1][e].cpu(), torch.from_numpy(ref[1][e]))
        assert torch.equal(out[3][e].cpu(), torch.from_numpy(ref[3][e]))
        assert out[5][e] == ref[5][e]
    for t in ('a', 'b', 'c'):
        assert torch.equal(out[2][t].cpu(), torch.from_numpy(ref[2][t]))
        assert out[4][t] == ref[4][t]


@pytest.mark.parametrize('level', ['node', 'edge'])
@pytest.mark.parametrize('strategy', ['uniform', 'last'])
@pytest.mark.parametrize('replace', [False, True])
def test_temporal_random_graph_matches_oracle(level, strategy, replace):
    # neighbourhoods sorted by time, as node_/edge_temporal_sample require (neighbor_kernel.cpp:74-144)
    rng = np.random.default_rng(8)
    n = 3000
    deg = rng.poisson(14, n).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    node_time = rng.integers(0, 1000, n, dtype=np.int64)
    edge_time = rng.integers(0, 1000, col.size, dtype=np.int64)
    for v in range(n):
        a, b = rowptr[v], rowptr[v + 1]
        if level == 'node':
            order = np.argsort(node_time[col[a:b]], kind='stable')
            col[a:b] = col[a:b][order]
        else:
            edge_time[a:b] = np.sort(edge_time[a:b])
    seeds = rng.permutation(n)[:48]
    kw = dict(disjoint=True, replace=replace, temporal_strategy=strategy)
    if level == 'node':
        kw['node_time'] = node_time
        if strategy == 'last':
            kw['seed_time'] = rng.integers(200, 1000, 48, dtype=np.int64)
    else:
        kw['edge_time'] = edge_time
        kw['seed_time'] = rng.integers(200, 1000, 48, dtype=np.int64)
    out, after, ref = run_both(rowptr, col, seeds, [6, 4, 3], 77, **kw)
    assert_same(out, after, ref, 77)
    assert sum(ref[5]) > 500


def test_temporal_unsorted_neighbourhood_raises():
    rowptr = np.array([0, 3, 3, 3, 3], dtype=np.int64)
    col = np.array([1, 2, 3], dtype=np.int64)
    node_time = np.array([9, 5, 1, 3], dtype=np.int64)  # times of the neighbours: 5, 1, 3 -> not sorted
    with pytest.raises(RuntimeError, match='non-sorted temporal'):
        sampler.neighbor_sample(dev(rowptr), dev(col), dev([0]), [2], node_time=dev(node_time), disjoint=True)


def test_int32_graphs_match_int64_and_keep_their_dtype():
    """The reference dispatches on the seeds' integral type (neighbor_kernel.cpp:893,930): int32 graphs give the
    same samples, the same generator advance, and int32 outputs."""
    rowptr, col = random_csr(3000, 12, 4)
    seed = np.arange(0, 300, 3, dtype=np.int64)
    for kw in (dict(), dict(replace=True), dict(disjoint=True)):
        torch.manual_seed(77)
        ref = sampler.neighbor_sample(dev(rowptr), dev(col), dev(seed), [6, 3], **kw)
        a64 = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
        torch.manual_seed(77)
        out = sampler.neighbor_sample(dev(rowptr).int(), dev(col).int(), dev(seed).int(), [6, 3], **kw)
        a32 = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
        assert a32 == a64
        for o, r in zip(out[:4], ref[:4]):
            assert o.dtype == torch.int32
            assert torch.equal(o.long(), r)
        assert out[4] == ref[4] and out[5] == ref[5]
    et = ('n', 'to', 'n')
    torch.manual_seed(5)
    h64 = sampler.hetero_neighbor_sample({et: dev(rowptr)}, {et: dev(col)}, {'n': dev(seed)}, {et: [4, 2]})
    torch.manual_seed(5)
    h32 = sampler.hetero_neighbor_sample({et: dev(rowptr).int()}, {et: dev(col).int()}, {'n': dev(seed).int()}, {et: [4, 2]})
    assert h32[0][et].dtype == torch.int32 and torch.equal(h32[0][et].long(), h64[0][et])
    assert torch.equal(h32[2]['n'].long(), h64[2]['n']) and torch.equal(h32[3][et].long(), h64[3][et])
    with pytest.raises(RuntimeError, match="seeds' dtype"):
        sampler.neighbor_sample(dev(rowptr), dev(col).int(), dev(seed), [2])


def test_unsupported_modes_fail_loudly():
    rowptr, col = dev(G.ROWPTR), dev(G.COL)
    with pytest.raises(RuntimeError):
        sampler.neighbor_sample(rowptr, col, dev([2, 3]), [2], directed=False)
    with pytest.raises(RuntimeError, match='disjoint'):
        sampler.neighbor_sample(rowptr, col, dev([2, 3]), [2], node_time=dev(np.arange(6)))
    with pytest.raises(RuntimeError, match='float32 or float64'):  # biased sampling itself: tests/test_biased_sampler_gpu.py
        sampler.neighbor_sample(rowptr, col, dev([2, 3]), [1], edge_weight=torch.ones(12, device=DEV).half())
    # a graph split across devices is refused by the dispatcher (CPU graphs themselves run the CPU kernel,
    # tests/test_cpu_key.py; a device graph never falls back to it)
    with pytest.raises(RuntimeError):
        sampler.neighbor_sample(rowptr, col, torch.tensor([2, 3]), [2])


def test_products_scale_batch_is_bit_exact():
    """BASELINE config C3 shape: N=2,449,029 nodes, log-normal degrees clipped to [1, 17481]
    (~100 M edges), fan-out [15, 10, 5], batch 1024, torch.manual_seed(12345)."""
    rng = np.random.default_rng(0)
    n = 2_449_029
    deg = np.clip(np.rint(rng.lognormal(3.3, 1.0, n)), 1, 17481).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    seeds = rng.permutation(n)[:1024].astype(np.int64)
    out, after, ref = run_both(rowptr, col, seeds, [15, 10, 5], 12345)
    assert sampler.last_mode() == 'fused'   # 2 - 3 launches per hop (csrc/hip/sampler_fused.h)
    assert_same(out, after, ref, 12345)
    assert sum(ref[5]) > 500_000


@pytest.mark.parametrize('num_seeds,disjoint', [(300, False), (1024, True), (3000, False), (3000, True), (70_000, False),
                                                (300_000, False)])
def test_every_scan_form_of_the_fused_chain(num_seeds, disjoint):
    """The seeds' and the hops' first-occurrence scans run in one of three forms (sampler_fused.h): seeds of one tile are
    inserted and scanned by ONE single-block launch together with the call's initialisation; scans of up to 256 tiles are
    one launch whose blocks look back at the aggregates in front of them; larger ones are a reduce + apply pair.  Seed
    counts on either side of both limits, repeated seeds (the duplicate-seed quirk of Mapper) included."""
    n = 400_000
    rowptr, col = random_csr(n, 8, seed=21)
    rng = np.random.default_rng(22)
    seeds = rng.integers(0, n, num_seeds, dtype=np.int64)
    seeds[num_seeds // 3] = seeds[0]                     # repeated seeds
    seeds[num_seeds // 2:num_seeds // 2 + 5] = seeds[1]
    out, after, ref = run_both(rowptr, col, seeds, [4, 3], 4242, disjoint=disjoint)
    assert sampler.last_mode() == 'fused'
    assert_same(out, after, ref, 4242)
    assert sum(ref[5]) > 3 * num_seeds


@pytest.mark.parametrize('case', G.DIST_CASES, ids=[c['name'] for c in G.DIST_CASES])
def test_dist_reference_golden_vectors(case):
    # test/csrc/sampler/test_dist_neighbor.cpp
    kw = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in case['kwargs'].items()}
    torch.manual_seed(case.get('manual_seed', 0))
    node, edge, cumsum = torch.ops.pyg.dist_neighbor_sample(dev(G.ROWPTR), dev(case['col']), dev(case['seed']),
                                                            case['num_neighbors'], **kw)
    assert node.cpu().tolist() == case['node']
    assert edge.cpu().tolist() == case['edge']
    assert cumsum == case['cumsum']


@pytest.mark.parametrize('variant', ['plain', 'replace', 'disjoint', 'full', 'temporal'])
def test_dist_random_graph_matches_oracle(variant):
    rowptr, col = random_csr(4000, 20, seed=12)
    seeds = np.random.default_rng(13).permutation(4000)[:500]
    kw, fan = {}, 7
    if variant == 'replace':
        kw = dict(replace=True)
    elif variant == 'disjoint':
        kw = dict(disjoint=True)
    elif variant == 'full':
        fan = -1
    elif variant == 'temporal':
        rng = np.random.default_rng(14)
        et = rng.integers(0, 500, col.size, dtype=np.int64)
        for v in range(4000):
            et[rowptr[v]:rowptr[v + 1]] = np.sort(et[rowptr[v]:rowptr[v + 1]])
        kw = dict(disjoint=True, edge_time=et, seed_time=rng.integers(100, 500, 500, dtype=np.int64))
    torch.manual_seed(5)
    dkw = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    node, edge, cumsum = torch.ops.pyg.dist_neighbor_sample(dev(rowptr), dev(col), dev(seeds), fan, **dkw)
    after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
    rnode, redge, rcumsum, info = oracle.dist_neighbor_sample(rowptr, col, seeds, fan, rng_seed=5, **kw)
    assert torch.equal(node.cpu(), torch.from_numpy(rnode))
    assert torch.equal(edge.cpu(), torch.from_numpy(redge))
    assert cumsum == rcumsum
    assert after == int(oracle.mt19937_words(5, info['rng_blocks'] * 128 + 1)[-1])
    # an int32 graph is read in place (the reference dispatches on the seeds' integral type, neighbor_kernel.cpp:893):
    # same draws, int32 results
    torch.manual_seed(5)
    n32, e32, c32 = torch.ops.pyg.dist_neighbor_sample(dev(rowptr).int(), dev(col).int(), dev(seeds).int(), fan, **dkw)
    assert n32.dtype == torch.int32 and e32.dtype == torch.int32
    assert torch.equal(n32.long(), node) and torch.equal(e32.long(), edge) and c32 == cumsum


def test_hetero_hub_forces_word_top_up_and_requeue():
    # A hub row of degree >= 2^16 makes its draws 32 bits wide: the first relation of the hop needs more
    # random words than the 16-bit speculation provided, raises the sticky abort, is repeated after a
    # top-up, and the relations queued behind it in the same hop start over (sampler.hip, commit()).
    rng = np.random.default_rng(12)
    na, nb = 1200, 70_000
    ets = [('a', 'r1', 'b'), ('a', 'r2', 'a'), ('b', 'r3', 'a'), ('b', 'r4', 'b')]
    deg = rng.poisson(6, na).astype(np.int64)
    deg[7] = 69_000  # the hub
    rp1 = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    cl1 = rng.integers(0, nb, int(rp1[-1]), dtype=np.int64)
    cl1[rp1[7]:rp1[8]] = rng.permutation(nb)[:69_000]
    rp, cl = {ets[0]: rp1}, {ets[0]: cl1}
    for et, (ns, nd) in zip(ets[1:], [(na, na), (nb, na), (nb, nb)]):
        d = rng.poisson(4, ns).astype(np.int64)
        rp[et] = np.concatenate([[0], np.cumsum(d)]).astype(np.int64)
        cl[et] = rng.integers(0, nd, int(d.sum()), dtype=np.int64)
    # 400 copies of the hub: 2400 32-bit draws = 1200 words, twice what the 16-bit speculation generates
    seeds = {'a': np.array([3, 7, 11, 500] + [7] * 400, dtype=np.int64), 'b': np.array([5, 69_999], dtype=np.int64)}
    fan = {e: [6, 3] for e in ets}
    fan[ets[1]] = [0, 3]  # keeps the hop's speculative chunk (868 words) below the 1212 the hub needs
    for replace in (False, True):
        torch.manual_seed(77)
        out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in rp.items()}, {e: dev(v) for e, v in cl.items()},
                                             {k: dev(v) for k, v in seeds.items()}, fan, replace=replace)
        after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
        ref = oracle.hetero_neighbor_sample(['a', 'b'], ets, rp, cl, seeds, fan, replace=replace, rng_seed=77)
        for e in ets:
            assert torch.equal(out[0][e].cpu(), torch.from_numpy(ref[0][e])), (e, replace)
            assert torch.equal(out[1][e].cpu(), torch.from_numpy(ref[1][e]))
            assert torch.equal(out[3][e].cpu(), torch.from_numpy(ref[3][e]))
        for t in ('a', 'b'):
            assert torch.equal(out[2][t].cpu(), torch.from_numpy(ref[2][t]))
        # the global CPU generator advanced by exactly the reference's number of 128-word prefetches
        assert after == int(oracle.mt19937_words(77, ref[6]['rng_blocks'] * 128 + 1)[-1])
        assert sum(sum(v) for v in ref[5].values()) > 100


def _table_cache(limit=0):
    import ctypes
    from pyg_lib_amd import _capi
    L = _capi.lib()
    L.pyg_hip_sampler_table_cache.argtypes = [ctypes.c_int64]
    L.pyg_hip_sampler_table_cache.restype = ctypes.c_int
    return L.pyg_hip_sampler_table_cache(limit)


def test_cached_node_tables_survive_reuse_other_graphs_and_epoch_wrap():
    """The fused chain keeps its direct-address node table between calls and invalidates old contents through an epoch in
    the value coding instead of clearing 8 bytes per node per call.  Every call below must stay bit-exact with the
    oracle: the same graph again (its old ids are stale now), OTHER graphs with the same node count (stale entries that
    name different nodes' positions), duplicate seeds, and -- with the epoch limit lowered to 3 -- the wrap-around, where
    the numbering restarts and the table is cleared once."""
    n = 20000
    graphs = [random_csr(n, 12, 50 + i) for i in range(3)]
    try:
        _table_cache(3)   # wrap after three calls per table
        for it in range(14):
            rowptr, col = graphs[it % 3]
            rng = np.random.default_rng(100 + it)
            seed = rng.choice(n, 300, replace=False)
            if it % 5 == 4:
                seed[7] = seed[3]
            out, after, ref = run_both(rowptr, col, seed, [10, 5, 3], 1000 + it)
            assert sampler.last_mode() == 'fused'
            assert_same(out, after, ref, 1000 + it)
        assert _table_cache(0) >= 1   # a table of this size is cached (and the limit is back at its default)
    finally:
        _table_cache(0)


def test_cached_node_tables_hetero_types_with_equal_node_counts():
    """Two node types with the SAME node count in one call need two cached tables (an entry serves one table at a time);
    repeated calls reuse both."""
    n = 6000
    types = ['a', 'b']
    ets = [('a', 'x', 'b'), ('b', 'y', 'a'), ('a', 'z', 'a')]
    rng = np.random.default_rng(7)
    rp, cl = {}, {}
    for i, et in enumerate(ets):
        rp[et], cl[et] = random_csr(n, 8, 70 + i)
    fan = {et: [6, 4] for et in ets}
    for it in range(5):
        seeds = {'a': rng.choice(n, 120, replace=False).astype(np.int64), 'b': rng.choice(n, 80, replace=False).astype(np.int64)}
        torch.manual_seed(77 + it)
        out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in rp.items()}, {e: dev(v) for e, v in cl.items()},
                                             {k: dev(v) for k, v in seeds.items()}, fan)
        assert sampler.last_mode() == 'fused'
        ref = oracle.hetero_neighbor_sample(types, ets, rp, cl, seeds, fan, rng_seed=77 + it)
        for e in ets:
            assert torch.equal(out[0][e].cpu(), torch.from_numpy(ref[0][e])), (it, e)
            assert torch.equal(out[1][e].cpu(), torch.from_numpy(ref[1][e])), (it, e)
            assert out[5][e] == ref[5][e]
        for t in types:
            assert torch.equal(out[2][t].cpu(), torch.from_numpy(ref[2][t])), (it, t)
    assert _table_cache(0) >= 2
