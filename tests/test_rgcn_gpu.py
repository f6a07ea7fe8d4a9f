"""End-to-end parity of the fused R-GCN step (BASELINE.json configs[4]):
hetero_neighbor_sample -> gather_coo -> segment_matmul -> scatter_sum on the device versus the
oracle sampler + a float64 numpy restatement of the same layer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def build_graph(rng, sizes, ets, mean_deg):
    rp, cl = {}, {}
    for (s, r, d) in ets:
        deg = rng.poisson(mean_deg, sizes[s]).astype(np.int64)
        rp[(s, r, d)] = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        cl[(s, r, d)] = rng.integers(0, sizes[d], int(deg.sum()), dtype=np.int64)
    return rp, cl


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_rgcn_layer_matches_oracle(dtype, tol):
    import oracle
    from pyg_lib_amd import sampler, rgcn
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(11)
    types = ['paper', 'author', 'inst']
    sizes = {'paper': 5000, 'author': 8000, 'inst': 300}
    ets = [('paper', 'cites', 'paper'), ('paper', 'rev_writes', 'author'), ('author', 'writes', 'paper'),
           ('author', 'affil', 'inst'), ('inst', 'rev_affil', 'author')]
    rp, cl = build_graph(rng, sizes, ets, 9)
    seeds = {'paper': rng.permutation(sizes['paper'])[:64].astype(np.int64)}
    fan = {e: [6, 4] for e in ets}
    F = 64
    feat = {t: rng.standard_normal((sizes[t], F)).astype(np.float32) for t in types}
    W = (rng.standard_normal((len(ets), F, F)) / np.sqrt(F)).astype(np.float32)

    torch.manual_seed(5)
    out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in rp.items()}, {e: dev(v) for e, v in cl.items()},
                                         {k: dev(v) for k, v in seeds.items()}, fan)
    row_d, col_d, node_d = out[0], out[1], out[2]
    ref = oracle.hetero_neighbor_sample(types, ets, rp, cl, seeds, fan, rng_seed=5)
    for t in types:
        assert torch.equal(node_d[t].cpu(), torch.from_numpy(ref[2][t]))

    nn = {t: node_d[t].numel() for t in types}
    off = rgcn.type_offsets(nn, types)
    featd = {t: dev(feat[t]).to(dtype) for t in types}
    x = torch.cat([featd[t][node_d[t]] for t in types])
    Wd = dev(W).to(dtype)
    y = rgcn.rgcn_layer(x, off, row_d, col_d, ets, Wd)
    assert y.shape == (off['__total__'], F)

    # float64 restatement on the oracle's sample, from the values the device actually holds
    xr = x.float().cpu().numpy().astype(np.float64)
    Wr = Wd.float().cpu().numpy().astype(np.float64)
    want = np.zeros((off['__total__'], F))
    for i, (s, r, d) in enumerate(ets):
        row, col = ref[0][(s, r, d)], ref[1][(s, r, d)]
        msg = xr[col + off[d]] @ Wr[i]
        if dtype != torch.float32:  # messages are rounded to the storage type before the reduction
            msg = torch.from_numpy(msg).to(dtype).double().numpy()
        np.add.at(want, row + off[s], msg)
    got = y.float().cpu().numpy().astype(np.float64)
    scale = np.abs(want).max()
    assert scale > 1.0
    assert np.abs(got - want).max() <= tol * scale
    # nodes that were never expanded receive nothing
    touched = np.zeros(off['__total__'], bool)
    for (s, r, d) in ets:
        touched[ref[0][(s, r, d)] + off[s]] = True
    assert not got[~touched].any()


def test_rgcn_layer_empty_sample():
    from pyg_lib_amd import rgcn
    ets = [('a', 'x', 'a')]
    x = torch.randn(4, 64, device='cuda')
    e = torch.zeros(0, dtype=torch.long, device='cuda')
    y = rgcn.rgcn_layer(x, rgcn.type_offsets({'a': 4}, ['a']), {ets[0]: e}, {ets[0]: e}, ets,
                        torch.randn(1, 64, 64, device='cuda'))
    assert y.shape == (4, 64) and not y.any()
