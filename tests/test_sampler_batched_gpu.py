"""K independent mini-batches in one call (pyg_hip_hetero_neighbor_sample_batched): every batch must be
bit for bit what the single-batch operator -- i.e. the oracle's restatement of sampler/cpu/neighbor_kernel.cpp:332-514 --
gives for that batch alone under torch.manual_seed(generator_seeds[b])."""
import numpy as np
import pytest
import torch

import oracle
from pyg_lib_amd import sampler

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def dev(a):
    return torch.as_tensor(np.asarray(a)).to(DEV)


def check_batch(out, ref):
    row, col, node, eid, nh, eh = out
    rrow, rcol, rnode, reid, rnh, reh, info = ref
    assert nh == rnh and eh == reh
    assert torch.equal(row.cpu(), torch.from_numpy(rrow)) and torch.equal(col.cpu(), torch.from_numpy(rcol))
    assert torch.equal(node.cpu(), torch.from_numpy(rnode))
    if eid is not None:
        assert torch.equal(eid.cpu(), torch.from_numpy(reid))


def test_c3_bench_graph_eight_batches_bit_exact_vs_oracle():
    """The graph bench_sampler.py times (synthetic ogbn-products scale: 2,449,029 nodes, log-normal(3.42) degrees, ~123 M
    edges; BASELINE config C3), K = 8 batches of 1024 seeds, fan-out [15, 10, 5], one generator seed per batch: every
    batch of the batched call against oracle.neighbor_sample on that batch alone, and against the single-batch operator
    (which must also have run the fused chain)."""
    import bench_sampler
    rowptr, col = bench_sampler.make_graph(DEV)
    rp, cl = rowptr.cpu().numpy(), col.cpu().numpy()
    g = torch.Generator().manual_seed(1)
    K = 8
    seeds = torch.randperm(bench_sampler.N_NODES, generator=g)[:K * bench_sampler.BATCH].view(K, -1)
    gseeds = [12345 + 7 * b for b in range(K)]
    before = torch.random.get_rng_state().clone()
    outs = sampler.neighbor_sample_batched(rowptr, col, [s.to(DEV) for s in seeds], bench_sampler.FANOUT, gseeds)
    assert torch.equal(torch.random.get_rng_state(), before)       # the default generator is not touched
    assert len(outs) == K
    for b in range(K):
        ref = oracle.neighbor_sample(rp, cl, seeds[b].numpy(), bench_sampler.FANOUT, rng_seed=gseeds[b])
        check_batch(outs[b], ref)
        assert sum(ref[5]) > 500_000
        if b in (0, 5):   # the single-batch operator on the bench graph (the C3 test graph was another one)
            torch.manual_seed(gseeds[b])
            one = sampler.neighbor_sample(rowptr, col, seeds[b].to(DEV), bench_sampler.FANOUT)
            assert sampler.last_mode() == 'fused'
            check_batch(one, ref)


@pytest.mark.parametrize('K', [1, 3, 8, 19])
@pytest.mark.parametrize('variant', ['plain', 'disjoint', 'replace', 'no_eid'])
def test_batched_equals_the_oracle_per_batch(K, variant):
    rng = np.random.default_rng(K)
    n = 60_000
    deg = rng.poisson(9, n).astype(np.int64)
    deg[rng.random(n) < 0.1] = 0
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    kw = dict(disjoint=variant == 'disjoint', replace=variant == 'replace', return_edge_id=variant != 'no_eid')
    sizes = [int(rng.integers(0, 700)) for _ in range(K)]
    sizes[0] = 333
    if K > 2:
        sizes[2] = 0                                  # an empty batch
    seeds = [rng.integers(0, n, s, dtype=np.int64) for s in sizes]
    gseeds = [int(v) for v in rng.integers(0, 2 ** 31, K)]
    outs = sampler.neighbor_sample_batched(dev(rowptr), dev(col), [dev(s) for s in seeds], [6, 4, 3], gseeds, **kw)
    for b in range(K):
        ref = oracle.neighbor_sample(rowptr, col, seeds[b], [6, 4, 3], rng_seed=gseeds[b], **kw)
        check_batch(outs[b], ref)
    # twice the same call: the same bits (no state carried between batched calls)
    again = sampler.neighbor_sample_batched(dev(rowptr), dev(col), [dev(s) for s in seeds], [6, 4, 3], gseeds, **kw)
    for a, o in zip(again, outs):
        assert torch.equal(a[0], o[0]) and torch.equal(a[1], o[1]) and torch.equal(a[2], o[2]) and a[4] == o[4] and a[5] == o[5]


def test_batched_int32_graph_and_argument_errors():
    rng = np.random.default_rng(5)
    n = 5000
    deg = rng.poisson(6, n).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.integers(0, n, int(rowptr[-1]), dtype=np.int64)
    seeds = [rng.integers(0, n, 100, dtype=np.int64) for _ in range(4)]
    o64 = sampler.neighbor_sample_batched(dev(rowptr), dev(col), [dev(s) for s in seeds], [5, 5], [1, 2, 3, 4])
    o32 = sampler.neighbor_sample_batched(dev(rowptr).int(), dev(col).int(), [dev(s).int() for s in seeds], [5, 5], [1, 2, 3, 4])
    for a, b in zip(o32, o64):
        assert a[0].dtype == torch.int32 and torch.equal(a[0].long(), b[0]) and torch.equal(a[2].long(), b[2]) and a[5] == b[5]
    with pytest.raises(RuntimeError, match='one generator seed per batch'):
        sampler.neighbor_sample_batched(dev(rowptr), dev(col), [dev(s) for s in seeds], [5, 5], [1, 2])
    with pytest.raises(RuntimeError):
        sampler.neighbor_sample_batched(dev(rowptr), dev(col), [dev(s) for s in seeds], [5, 5], [1, 2, 3, 4], directed=False)
    assert sampler.neighbor_sample_batched(dev(rowptr), dev(col), [], [5, 5], []) == []


def test_hetero_batched_equals_single_calls():
    """MAG-shaped toy graph: K hetero batches in one call against K single hetero_neighbor_sample calls under
    torch.manual_seed (themselves pinned on the oracle in tests/test_rgcn_gpu.py / test_sampler_gpu.py)."""
    rng = np.random.default_rng(11)
    sizes = {'paper': 30_000, 'author': 20_000, 'inst': 900}
    ets = [('author', 'writes', 'paper'), ('paper', 'cites', 'paper'), ('paper', 'rev_writes', 'author'),
           ('author', 'at', 'inst'), ('inst', 'rev_at', 'author')]
    rowptr, col, fan = {}, {}, {}
    for et in ets:
        ns, nd = sizes[et[0]], sizes[et[2]]
        deg = rng.poisson(5, ns).astype(np.int64)
        rowptr[et] = dev(np.concatenate([[0], np.cumsum(deg)]).astype(np.int64))
        col[et] = dev(rng.integers(0, nd, int(deg.sum()), dtype=np.int64))
        fan[et] = [4, 3]
    K = 6
    seed_dicts = [{'paper': dev(rng.integers(0, sizes['paper'], 200, dtype=np.int64)),
                   'author': dev(rng.integers(0, sizes['author'], 50 + 10 * b, dtype=np.int64))} for b in range(K)]
    gseeds = [100 + b for b in range(K)]
    outs = sampler.hetero_neighbor_sample_batched(rowptr, col, seed_dicts, fan, gseeds)
    for b in range(K):
        torch.manual_seed(gseeds[b])
        one = sampler.hetero_neighbor_sample(rowptr, col, seed_dicts[b], fan)
        for et in ets:
            assert torch.equal(outs[b][0][et], one[0][et]) and torch.equal(outs[b][1][et], one[1][et])
            assert torch.equal(outs[b][3][et], one[3][et]) and outs[b][5][et] == one[5][et]
        for t in sizes:
            assert torch.equal(outs[b][2][t], one[2][t]) and outs[b][4][t] == one[4][t]


@pytest.mark.parametrize('mode', ['node_time', 'node_time_last', 'edge_time', 'weight'])
def test_hetero_batched_temporal_and_biased_equal_single_calls(mode):
    """The batched hetero entry takes every mode of hetero_neighbor_sample (the reference has one entry for all of them,
    sampler/neighbor.cpp:137-147): node-level / edge-level temporal sampling with one seed_time dict per batch, the 'last'
    strategy, biased sampling -- each batch bit for bit the single call under torch.manual_seed(generator_seeds[b]) (the
    single calls are pinned on the oracle in tests/test_sampler_gpu.py / test_biased_sampler_gpu.py)."""
    rng = np.random.default_rng(21)
    sizes = {'paper': 9_000, 'author': 6_000, 'inst': 300}
    ets = [('author', 'writes', 'paper'), ('paper', 'cites', 'paper'), ('paper', 'rev_writes', 'author'),
           ('author', 'at', 'inst'), ('inst', 'rev_at', 'author')]
    rowptr, col, fan, etime, weight = {}, {}, {}, {}, {}
    ntime_np = {t: rng.integers(0, 1000, n, dtype=np.int64) for t, n in sizes.items()}
    for et in ets:
        ns, nd = sizes[et[0]], sizes[et[2]]
        deg = rng.poisson(6, ns).astype(np.int64)
        rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        c = rng.integers(0, nd, int(deg.sum()), dtype=np.int64)
        te = rng.integers(0, 1000, int(deg.sum()), dtype=np.int64)
        rowid = np.repeat(np.arange(ns), deg)
        # the reference requires every neighbourhood sorted by time (sampler/cpu/neighbor_kernel.cpp: "Found invalid non-sorted
        # temporal neighborhood"): by the neighbour's node time / by the edge time
        if mode.startswith('node_time'):
            c = c[np.lexsort((ntime_np[et[2]][c], rowid))]
        te = te[np.lexsort((te, rowid))]
        rowptr[et] = dev(rp)
        col[et] = dev(c)
        etime[et] = dev(te)
        w = rng.random(int(deg.sum()))
        w[rng.random(w.size) < 0.2] = 0.0
        weight[et] = dev(w)
        fan[et] = [4, 3]
    ntime = {t: dev(v) for t, v in ntime_np.items()}
    K = 5
    seed_dicts = [{'paper': dev(rng.integers(0, sizes['paper'], 120, dtype=np.int64)),
                   'author': dev(rng.integers(0, sizes['author'], 30 + 7 * b, dtype=np.int64))} for b in range(K)]
    stimes = [{t: dev(rng.integers(200, 1000, v.numel(), dtype=np.int64)) for t, v in d.items()} for d in seed_dicts]
    gseeds = [500 + b for b in range(K)]
    if mode.startswith('node_time'):
        kw = dict(node_time_dict=ntime, disjoint=True, temporal_strategy='last' if mode.endswith('last') else 'uniform')
        per_batch = [dict(seed_time_dict=stimes[b]) for b in range(K)]
        outs = sampler.hetero_neighbor_sample_batched(rowptr, col, seed_dicts, fan, gseeds, seed_time_dicts=stimes, **kw)
    elif mode == 'edge_time':
        kw = dict(edge_time_dict=etime, disjoint=True)
        per_batch = [dict(seed_time_dict=stimes[b]) for b in range(K)]
        outs = sampler.hetero_neighbor_sample_batched(rowptr, col, seed_dicts, fan, gseeds, seed_time_dicts=stimes, **kw)
    else:
        kw = dict(edge_weight_dict=weight)
        per_batch = [dict() for _ in range(K)]
        outs = sampler.hetero_neighbor_sample_batched(rowptr, col, seed_dicts, fan, gseeds, **kw)
    edges = 0
    for b in range(K):
        torch.manual_seed(gseeds[b])
        one = sampler.hetero_neighbor_sample(rowptr, col, seed_dicts[b], fan, **kw, **per_batch[b])
        for et in ets:
            assert torch.equal(outs[b][0][et], one[0][et]) and torch.equal(outs[b][1][et], one[1][et])
            assert torch.equal(outs[b][3][et], one[3][et]) and outs[b][5][et] == one[5][et]
            edges += one[0][et].numel()
        for t in sizes:
            assert torch.equal(outs[b][2][t], one[2][t]) and outs[b][4][t] == one[4][t]
    assert edges > 1000
    with pytest.raises(RuntimeError, match='disjoint'):
        sampler.hetero_neighbor_sample_batched(rowptr, col, seed_dicts, fan, gseeds, node_time_dict=ntime)
    with pytest.raises(RuntimeError, match='one seed_time dict per batch'):
        sampler.hetero_neighbor_sample_batched(rowptr, col, seed_dicts, fan, gseeds, edge_time_dict=etime, disjoint=True,
                                               seed_time_dicts=stimes[:2])


def test_table_cache_release_returns_the_memory():
    rng = np.random.default_rng(3)
    n = 300_000
    deg = rng.poisson(6, n).astype(np.int64)
    rowptr = dev(np.concatenate([[0], np.cumsum(deg)]).astype(np.int64))
    col = dev(rng.integers(0, n, int(deg.sum()), dtype=np.int64))
    seeds = [dev(rng.integers(0, n, 500, dtype=np.int64)) for _ in range(4)]
    first = sampler.neighbor_sample_batched(rowptr, col, seeds, [5, 5], [1, 2, 3, 4])
    import ctypes
    from pyg_lib_amd import _capi
    L = _capi.lib()
    L.pyg_hip_sampler_table_cache.argtypes = [ctypes.c_int64]
    assert L.pyg_hip_sampler_table_cache(0) >= 1          # tables are kept between calls ...
    assert sampler.release_table_cache() == 0             # ... until released (none busy now)
    assert L.pyg_hip_sampler_table_cache(0) == 0
    again = sampler.neighbor_sample_batched(rowptr, col, seeds, [5, 5], [1, 2, 3, 4])   # and come back on demand
    for a, o in zip(again, first):
        assert torch.equal(a[0], o[0]) and torch.equal(a[2], o[2])
