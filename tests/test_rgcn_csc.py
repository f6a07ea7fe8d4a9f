"""``csc=True`` through the R-GCN chain on the CPU key (no GPU): the mode of the reference's own MAG benchmark
(benchmark/sampler/hetero_neighbor.py:106-124).  For edge type (src, rel, dst) the sampler then returns ``row`` = the
SAMPLED neighbours (src-typed local ids) and ``col`` = the EXPANDED nodes (dst-typed, nondecreasing)
(pyg_lib/csrc/sampler/cpu/neighbor_kernel.cpp:715-719 and :147-159), and messages flow row -> col:

    out[col + off[dst]] += x[row + off[src]] @ W_r

Node types of UNEQUAL sizes make any confusion of the two roles an out-of-range index or a wrong sum."""
import numpy as np
import pytest
import torch

import oracle
from pyg_lib_amd import rgcn, sampler


def random_csc(rng, n_ptr, n_val, mean_deg):
    deg = rng.poisson(mean_deg, n_ptr).astype(np.int64)
    return (np.concatenate([[0], np.cumsum(deg)]).astype(np.int64),
            rng.integers(0, n_val, int(deg.sum()), dtype=np.int64))


def restate(x, off, rows, cols, ets, W, csc):
    want = np.zeros((off['__total__'], W.shape[2]))
    for i, (s, _, d) in enumerate(ets):
        row, col = np.asarray(rows[(s, _, d)]), np.asarray(cols[(s, _, d)])
        if csc:
            np.add.at(want, col + off[d], x[row + off[s]] @ W[i])
        else:
            np.add.at(want, row + off[s], x[col + off[d]] @ W[i])
    return want


@pytest.mark.parametrize('csc', [True, False])
def test_chain_on_the_oracle_sample_unequal_type_sizes(csc):
    rng = np.random.default_rng(5)
    types = ['a', 'b', 'c']
    sizes = {'a': 7, 'b': 400, 'c': 90}
    ets = [('a', 'r0', 'b'), ('b', 'r1', 'a'), ('b', 'r2', 'c'), ('c', 'r3', 'b'), ('b', 'r4', 'b')]
    rp, cl = {}, {}
    for (s, r, d) in ets:
        p, v = (d, s) if csc else (s, d)   # csc: a CSC over the dst nodes holding src ids
        rp[(s, r, d)], cl[(s, r, d)] = random_csc(rng, sizes[p], sizes[v], 5)
    seeds = {'b': rng.permutation(sizes['b'])[:40].astype(np.int64)}
    fan = {e: [4, 3] for e in ets}
    torch.manual_seed(12)
    t = torch.from_numpy
    out = sampler.hetero_neighbor_sample({e: t(v) for e, v in rp.items()}, {e: t(v) for e, v in cl.items()},
                                         {k: t(v) for k, v in seeds.items()}, fan, csc=csc)
    ref = oracle.hetero_neighbor_sample(types, ets, rp, cl, seeds, fan, csc=csc, rng_seed=12)
    row_d, col_d, node_d = out[0], out[1], out[2]
    for e in ets:
        assert np.array_equal(row_d[e].numpy(), ref[0][e]) and np.array_equal(col_d[e].numpy(), ref[1][e])
    nn = {ty: node_d[ty].numel() for ty in types}
    assert len(set(nn.values())) == 3 and nn['a'] <= 7
    # the roles: the expanded vector is nondecreasing and typed by the CSR's pointer side
    for (s, r, d) in ets:
        exp, exp_t = (col_d, d) if csc else (row_d, s)
        smp, smp_t = (row_d, s) if csc else (col_d, d)
        e = exp[(s, r, d)]
        assert bool((e[1:] >= e[:-1]).all())
        if e.numel():
            assert int(e.max()) < nn[exp_t] and int(smp[(s, r, d)].max()) < nn[smp_t]
    off = rgcn.type_offsets(nn, types)
    F = 16
    x = torch.from_numpy(rng.integers(-4, 5, (off['__total__'], F)).astype(np.float32))
    W = torch.from_numpy(rng.integers(-2, 3, (len(ets), F, F)).astype(np.float32))
    y = rgcn.rgcn_layer(x, off, row_d, col_d, ets, W, csc=csc)
    want = restate(x.double().numpy(), off, ref[0], ref[1], ets, W.double().numpy(), csc)
    assert np.abs(want).max() > 8
    assert np.array_equal(y.double().numpy(), want)     # integer data: exact
    # the fused entry points resolve to the same chain on the CPU key
    assert torch.equal(rgcn.rgcn_layer_fused(x, off, row_d, col_d, ets, W, csc=csc), y)
    feat = {ty: torch.from_numpy(rng.integers(-4, 5, (sizes[ty], F)).astype(np.float32)) for ty in types}
    yt = rgcn.rgcn_layer_fused_tables(feat, node_d, types, row_d, col_d, ets, W, csc=csc)
    xt = torch.cat([feat[ty][node_d[ty]] for ty in types])
    assert np.array_equal(yt.double().numpy(), restate(xt.double().numpy(), off, ref[0], ref[1], ets, W.double().numpy(), csc))


def test_edge_roles():
    et = ('s', 'r', 'd')
    row, col = torch.tensor([1]), torch.tensor([2])
    assert rgcn.edge_roles(et, {et: row}, {et: col}, csc=False) == (col, 'd', row, 's')
    assert rgcn.edge_roles(et, {et: row}, {et: col}, csc=True) == (row, 's', col, 'd')


@pytest.mark.parametrize('csc', [True, False])
def test_num_out_rows_trims_the_output_to_the_expanded_nodes(csc):
    """`num_out_rows` = the reductions' dim_size per node type (pyg_lib/csrc/ops/scatter.cpp:156-160): rows behind it are
    not produced; the rows in front of it equal the untrimmed result bit for bit (CPU key: the chain)."""
    rng = np.random.default_rng(9)
    types = ['a', 'b', 'c']
    sizes = {'a': 30, 'b': 400, 'c': 90}
    ets = [('a', 'r0', 'b'), ('b', 'r1', 'a'), ('b', 'r2', 'c'), ('c', 'r3', 'b'), ('b', 'r4', 'b')]
    rp, cl = {}, {}
    for (s, r, d) in ets:
        p, v = (d, s) if csc else (s, d)
        rp[(s, r, d)], cl[(s, r, d)] = random_csc(rng, sizes[p], sizes[v], 5)
    t = torch.from_numpy
    torch.manual_seed(3)
    out = sampler.hetero_neighbor_sample({e: t(v) for e, v in rp.items()}, {e: t(v) for e, v in cl.items()},
                                         {'b': t(rng.permutation(sizes['b'])[:20].astype(np.int64))}, {e: [3, 2] for e in ets}, csc=csc)
    row_d, col_d, node_d, nph = out[0], out[1], out[2], out[4]
    nn = {ty: node_d[ty].numel() for ty in types}
    off = rgcn.type_offsets(nn, types)
    expanded = {ty: int(sum(nph[ty][:-1])) for ty in types}   # the last hop's discoveries are never expanded
    assert sum(expanded.values()) < sum(nn.values())
    F = 8
    x = torch.from_numpy(rng.integers(-4, 5, (off['__total__'], F)).astype(np.float32))
    W = torch.from_numpy(rng.integers(-2, 3, (len(ets), F, F)).astype(np.float32))
    full = rgcn.rgcn_layer(x, off, row_d, col_d, ets, W, csc=csc)
    trim = rgcn.rgcn_layer(x, off, row_d, col_d, ets, W, csc=csc, num_out_rows=expanded)
    ooff = rgcn.out_offsets(off, expanded)
    assert trim.shape == (sum(expanded.values()), F) and ooff['__total__'] == trim.size(0)
    for ty in types:
        assert torch.equal(trim[ooff[ty]:ooff[ty] + expanded[ty]], full[off[ty]:off[ty] + expanded[ty]])
        assert not full[off[ty] + expanded[ty]:off[ty] + nn[ty]].any()   # what the trimmed form leaves out is zero rows
    assert torch.equal(rgcn.rgcn_layer_fused(x, off, row_d, col_d, ets, W, csc=csc, num_out_rows=expanded), trim)
    # a count that is too small is an error, as a dim_size that is too small is for the reductions
    small = dict(expanded)
    small['b'] = 1
    with pytest.raises(RuntimeError, match='num_out_rows'):
        rgcn.rgcn_layer(x, off, row_d, col_d, ets, W, csc=csc, num_out_rows=small)
    with pytest.raises(ValueError, match='unknown node type'):
        rgcn.out_offsets(off, {'zzz': 3})
