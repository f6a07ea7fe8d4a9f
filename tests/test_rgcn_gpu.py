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


# ---- fused layer (csrc/hip/rgcn.hip) -----------------------------------------------------------------------------

MAG_TYPES = ['paper', 'author', 'institution', 'field_of_study']
MAG_ETS = [('paper', 'cites', 'paper'), ('author', 'writes', 'paper'), ('paper', 'rev_writes', 'author'),
           ('author', 'affiliated_with', 'institution'), ('institution', 'rev_affiliated_with', 'author'),
           ('paper', 'has_topic', 'field_of_study'), ('field_of_study', 'rev_has_topic', 'paper')]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_fused_rgcn_layer_mag_shape_matches_oracle(dtype):
    """BASELINE config C5's shape in small: 4 node types, the 7 relations of ogbn-mag after ToUndirected, F = 128,
    batch 1024 papers, fan-out [15, 10] -- sampled by the device sampler (checked against the oracle sampler), layer by
    pyg::rgcn_fused, compared with a float64 restatement on the ORACLE's sample and with the three-op chain."""
    import oracle
    from pyg_lib_amd import sampler, rgcn
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(3)
    sizes = {'paper': 40_000, 'author': 60_000, 'institution': 900, 'field_of_study': 4_000}
    rp, cl = build_graph(rng, sizes, MAG_ETS, 12)
    seeds = {'paper': rng.permutation(sizes['paper'])[:1024].astype(np.int64)}
    fan = {e: [15, 10] for e in MAG_ETS}
    F = 128
    feat = {t: rng.standard_normal((sizes[t], F)).astype(np.float32) for t in MAG_TYPES}
    W = (rng.standard_normal((len(MAG_ETS), F, F)) / np.sqrt(F)).astype(np.float32)

    torch.manual_seed(9)
    out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in rp.items()}, {e: dev(v) for e, v in cl.items()},
                                         {k: dev(v) for k, v in seeds.items()}, fan)
    row_d, col_d, node_d = out[0], out[1], out[2]
    ref = oracle.hetero_neighbor_sample(MAG_TYPES, MAG_ETS, rp, cl, seeds, fan, rng_seed=9)
    for t in MAG_TYPES:
        assert torch.equal(node_d[t].cpu(), torch.from_numpy(ref[2][t]))
    for e in MAG_ETS:
        assert torch.equal(row_d[e].cpu(), torch.from_numpy(ref[0][e])) and torch.equal(col_d[e].cpu(), torch.from_numpy(ref[1][e]))
    off = rgcn.type_offsets({t: node_d[t].numel() for t in MAG_TYPES}, MAG_TYPES)
    featd = {t: dev(feat[t]).to(dtype) for t in MAG_TYPES}
    x = torch.cat([featd[t][node_d[t]] for t in MAG_TYPES])
    Wd = dev(W).to(dtype)
    y = rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, Wd, grouped=False)   # the atomic kernel (sampler rows would default to the atomic-free one)
    assert y.shape == (off['__total__'], F) and y.dtype == dtype
    total_edges = sum(v.numel() for v in row_d.values())
    assert total_edges > 100_000

    xr = x.float().cpu().numpy().astype(np.float64)
    Wr = Wd.float().cpu().numpy().astype(np.float64)
    want = np.zeros((off['__total__'], F))
    for i, (s, r, d) in enumerate(MAG_ETS):
        row, col = ref[0][(s, r, d)], ref[1][(s, r, d)]
        msg = torch.from_numpy(xr[col + off[d]] @ Wr[i]).to(dtype).double().numpy()  # messages rounded once to T
        np.add.at(want, row + off[s], msg)
    got = y.float().cpu().numpy().astype(np.float64)
    scale = np.abs(want).max()
    assert scale > 1.0
    # every destination row is a sum of <= 15 + ... messages over <= 4 relations, each partial sum rounded once
    assert np.abs(got - want).max() <= (2e-2 if dtype == torch.bfloat16 else 3e-3) * scale
    touched = np.zeros(off['__total__'], bool)
    for e in MAG_ETS:
        touched[ref[0][e] + off[e[0]]] = True
    assert not got[~touched].any()
    # the three-op chain agrees to the same tolerance
    y3 = rgcn.rgcn_layer(x, off, row_d, col_d, MAG_ETS, Wd).float().cpu().numpy()
    assert np.abs(got - y3).max() <= (3e-2 if dtype == torch.bfloat16 else 4e-3) * scale


def test_fused_rgcn_integer_valued_inputs_are_exact():
    """Small integers times signed-permutation weights: every message and every partial sum is exactly representable,
    so the fused layer must equal the float64 result bit for bit -- runs straddling tile and wave boundaries, a
    relation with fewer than 32 edges, an empty relation, destinations shared by several relations."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(2)
    n, F = 700, 128
    x = torch.randint(-4, 5, (n, F), generator=g).float()
    counts = [1000, 0, 17, 4096 + 33, 129]
    ets = [('a', f'r{i}', 'a') for i in range(len(counts))]
    perm = torch.stack([torch.randperm(F, generator=g) for _ in counts])
    W = torch.zeros(len(counts), F, F)
    W[torch.arange(len(counts))[:, None], perm, torch.arange(F)[None, :]] = (torch.randint(0, 2, (len(counts), F), generator=g) * 2 - 1).float()
    rows, cols = {}, {}
    for et, c in zip(ets, counts):
        r = torch.sort(torch.randint(0, 60, (c,), generator=g)).values  # long runs, few destinations
        rows[et] = r.cuda()
        cols[et] = torch.randint(0, n, (c,), generator=g).cuda()
    off = rgcn.type_offsets({'a': n}, ['a'])
    y = rgcn.rgcn_layer_fused(x.bfloat16().cuda(), off, rows, cols, ets, W.bfloat16().cuda())
    want = torch.zeros(n, F, dtype=torch.float64)
    for i, et in enumerate(ets):
        want.index_add_(0, rows[et].cpu(), x[cols[et].cpu()].double() @ W[i].double())
    assert want.abs().max() <= 256  # exactly representable in bf16
    assert torch.equal(y.double().cpu(), want)


def test_fused_rgcn_more_relations_than_the_kernel_argument_holds():
    """Up to 24 relation records travel in the kernel argument; longer lists are staged in the workspace (rgcn.hip,
    RgcnDesc).  30 relations (some empty, some shorter than a wave's 32 edges), integer data, exact result."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(13)
    n, F, R = 900, 128, 30
    x = torch.randint(-3, 4, (n, F), generator=g).float()
    counts = [int(c) for c in torch.randint(0, 700, (R,), generator=g)]
    counts[3] = 0
    counts[7] = 5
    counts[29] = 1300
    ets = [('a', f'r{i}', 'a') for i in range(R)]
    perm = torch.stack([torch.randperm(F, generator=g) for _ in range(R)])
    W = torch.zeros(R, F, F)
    W[torch.arange(R)[:, None], perm, torch.arange(F)[None, :]] = (torch.randint(0, 2, (R, F), generator=g) * 2 - 1).float()
    rows, cols = {}, {}
    for et, c in zip(ets, counts):
        rows[et] = torch.sort(torch.randint(0, 50, (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, n, (c,), generator=g).cuda()
    off = rgcn.type_offsets({'a': n}, ['a'])
    y = rgcn.rgcn_layer_fused(x.bfloat16().cuda(), off, rows, cols, ets, W.bfloat16().cuda())
    want = torch.zeros(n, F, dtype=torch.float64)
    for i, et in enumerate(ets):
        want.index_add_(0, rows[et].cpu(), x[cols[et].cpu()].double() @ W[i].double())
    assert want.abs().max() <= 256
    assert torch.equal(y.double().cpu(), want)


def test_fused_rgcn_feature_table_of_more_than_4_gib():
    """Row offsets need 64 bits once a feature table reaches 4 GiB (2^24 rows of 256 bytes): the kernel is instantiated
    for both widths (rgcn.hip, BIG).  Rows on either side of the 4 GiB line are gathered -- directly and through a node-id
    map -- from a table that is only initialised where it is read; integer data, exact result."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(9)
    F = 128
    n_big = (1 << 24) + 4096
    free, _ = torch.cuda.mem_get_info()
    if free < 3 * n_big * F * 2:
        pytest.skip('not enough device memory for a 4 GiB table')
    picks = torch.cat([1 + 50_000 * torch.randperm(300, generator=g), (1 << 24) + 1 + torch.randperm(4000, generator=g)[:300],
                       torch.tensor([0, (1 << 24) - 1, 1 << 24, n_big - 1])])  # distinct rows on both sides of 4 GiB
    assert picks.unique().numel() == picks.numel()
    table = torch.empty(n_big, F, dtype=torch.bfloat16, device='cuda')
    vals = torch.randint(-4, 5, (picks.numel(), F), generator=g).float()
    table[picks.cuda()] = vals.bfloat16().cuda()
    ets = [('a', 'r0', 'a'), ('a', 'r1', 'a')]
    counts = [5000, 77]
    perm = torch.stack([torch.randperm(F, generator=g) for _ in ets])
    W = torch.zeros(len(ets), F, F)
    W[torch.arange(len(ets))[:, None], perm, torch.arange(F)[None, :]] = (torch.randint(0, 2, (len(ets), F), generator=g) * 2 - 1).float()
    rows, cols = {}, {}
    for et, c in zip(ets, counts):
        rows[et] = torch.sort(torch.randint(0, 40, (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, picks.numel(), (c,), generator=g).cuda()
    # (a) through the node-id map: local node i is global row picks[i]
    y = rgcn.rgcn_layer_fused_tables({'a': table}, {'a': picks.cuda()}, ['a'], rows, cols, ets, W.bfloat16().cuda())
    want = torch.zeros(picks.numel(), F, dtype=torch.float64)
    for i, et in enumerate(ets):
        want.index_add_(0, rows[et].cpu(), vals[cols[et].cpu()].double() @ W[i].double())
    assert want.abs().max() <= 256
    assert torch.equal(y.double().cpu(), want)
    # (b) directly: x is the big table itself, the gather index the global row
    off = rgcn.type_offsets({'a': n_big}, ['a'])
    cols_g = {et: picks.cuda()[cols[et]] for et in ets}
    y2 = rgcn.rgcn_layer_fused(table, off, rows, cols_g, ets, W.bfloat16().cuda())
    assert y2.shape == (n_big, F)
    assert torch.equal(y2[:picks.numel()].double().cpu(), want)
    del table, y2


def test_fused_rgcn_falls_back_for_other_shapes():
    from pyg_lib_amd import rgcn
    ets = [('a', 'x', 'a')]
    x = torch.randn(50, 64, device='cuda')
    r = torch.sort(torch.randint(0, 50, (300,), device='cuda')).values
    c = torch.randint(0, 50, (300,), device='cuda')
    w = torch.randn(1, 64, 64, device='cuda')
    off = rgcn.type_offsets({'a': 50}, ['a'])
    torch.testing.assert_close(rgcn.rgcn_layer_fused(x, off, {ets[0]: r}, {ets[0]: c}, ets, w),
                               rgcn.rgcn_layer(x, off, {ets[0]: r}, {ets[0]: c}, ets, w))


def test_full_size_c5_hetero_sample_is_bit_exact_and_layer_matches():
    """BASELINE config C5 at FULL size (the graph bench.py's `c5` leg samples: 1.94 M nodes of 4 types, the 7 relations,
    42.2 M entries; batch 1024 papers, fan-out [15, 10]): every output of the device sampler -- rows, cols, node ids,
    edge ids, per-hop counts of all relations / node types -- against the oracle on one batch, then the fused layer on
    that sample against a float64 restatement."""
    import oracle
    import bench_legs
    from pyg_lib_amd import sampler, rgcn
    types = list(bench_legs.MAG_SIZES)
    ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
    rp, cl = bench_legs.make_mag_graph(torch.device('cuda:0'))
    fan = {e: [15, 10] for e in ets}
    seeds = torch.randperm(bench_legs.MAG_SIZES['paper'], generator=torch.Generator().manual_seed(1))[:1024]
    torch.manual_seed(2024)
    out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds.cuda()}, fan)
    assert sampler.last_mode() == 'fused'   # every relation of a hop in one launch per phase (sampler_fused.h)
    after = torch.get_rng_state()
    ref = oracle.hetero_neighbor_sample(types, ets, {e: v.cpu().numpy() for e, v in rp.items()},
                                        {e: v.cpu().numpy() for e, v in cl.items()}, {'paper': seeds.numpy()}, fan,
                                        rng_seed=2024)
    row_d, col_d, node_d, edge_d = out[0], out[1], out[2], out[3]
    total = 0
    for e in ets:
        assert torch.equal(row_d[e].cpu(), torch.from_numpy(ref[0][e])), e
        assert torch.equal(col_d[e].cpu(), torch.from_numpy(ref[1][e])), e
        assert torch.equal(edge_d[e].cpu(), torch.from_numpy(ref[3][e])), e
        assert list(out[5][e]) == list(ref[5][e]), e
        total += row_d[e].numel()
    for t in types:
        assert torch.equal(node_d[t].cpu(), torch.from_numpy(ref[2][t])), t
        assert list(out[4][t]) == list(ref[4][t]), t
    assert total > 400_000
    # the generator was advanced exactly as the reference's engine would have advanced it
    torch.manual_seed(2024)
    sampler.hetero_neighbor_sample({e: v.cpu() for e, v in rp.items()}, {e: v.cpu() for e, v in cl.items()},
                                   {'paper': seeds}, fan)  # the CPU key of this build: same draws from the generator
    assert torch.equal(after, torch.get_rng_state())

    F = 128
    g = torch.Generator(device='cuda').manual_seed(3)
    feat = {t: torch.randn(bench_legs.MAG_SIZES[t], F, device='cuda', generator=g).bfloat16() for t in types}
    W = (torch.randn(len(ets), F, F, device='cuda', generator=g) / F ** 0.5).bfloat16()
    off = rgcn.type_offsets({t: node_d[t].numel() for t in types}, types)
    x = torch.cat([feat[t][node_d[t]] for t in types])
    y = rgcn.rgcn_layer_fused(x, off, row_d, col_d, ets, W, grouped=False)   # the atomic kernel
    want = torch.zeros(off['__total__'], F, dtype=torch.float64, device='cuda')
    for i, (s, r, d) in enumerate(ets):
        msg = (x[col_d[(s, r, d)] + off[d]].double() @ W[i].double()).bfloat16().double()
        want.index_add_(0, row_d[(s, r, d)] + off[s], msg)
    scale = want.abs().max().item()
    assert scale > 1.0 and (y.double() - want).abs().max().item() <= 2e-2 * scale
    # the variant bench_legs.leg_c5 TIMES: rows gathered from the global feature tables through the sampler's node ids
    # inside the kernel (no per-batch x) -- same float64 restatement, same bar, and close to the materialised form
    # (only the 2 k-node case had this check)
    yt = rgcn.rgcn_layer_fused_tables(feat, node_d, types, row_d, col_d, ets, W, grouped=False)
    assert yt.shape == y.shape
    assert (yt.double() - want).abs().max().item() <= 2e-2 * scale
    assert (yt.double() - y.double()).abs().max().item() <= 2e-2 * scale   # (atomic order differs between two launches)
    # ... and the atomic-free kernel bench_legs.leg_c5 times since round 5 (grouped=True: the sampler's rows are
    # nondecreasing per relation, verified on the device): same bar, the same bits on a second run
    yg = rgcn.rgcn_layer_fused_tables(feat, node_d, types, row_d, col_d, ets, W, grouped=True)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    assert (yg.double() - want).abs().max().item() <= 2e-2 * scale
    assert torch.equal(yg, rgcn.rgcn_layer_fused_tables(feat, node_d, types, row_d, col_d, ets, W, grouped=True))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_fused_tables_equals_fused_on_the_materialised_features(dtype):
    """pyg::rgcn_fused_tables gathers feat[type][node_id[type][col]] inside the kernel: it must produce the very bits
    of pyg::rgcn_fused on x = cat(feat[t][node_id[t]]) when the features are integer valued (exact sums), and agree
    within one rounding per run otherwise."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(5)
    types = ['a', 'b', 'c']
    n_global = {'a': 5000, 'b': 300, 'c': 20000}
    n_local = {'a': 700, 'b': 64, 'c': 1500}
    ets = [('a', 'r0', 'a'), ('a', 'r1', 'b'), ('b', 'r2', 'a'), ('c', 'r3', 'a'), ('a', 'r4', 'c'), ('c', 'r5', 'c')]
    counts = [4000, 33, 1000, 0, 129, 9000]
    F = 128
    for integer in (True, False):
        feat = {t: (torch.randint(-3, 4, (n_global[t], F), generator=g).float() if integer
                    else torch.randn(n_global[t], F, generator=g)).to(dtype).cuda() for t in types}
        node_id = {t: torch.randperm(n_global[t], generator=g)[:n_local[t]].cuda() for t in types}
        if integer:
            perm = torch.stack([torch.randperm(F, generator=g) for _ in ets])
            W = torch.zeros(len(ets), F, F)
            W[torch.arange(len(ets))[:, None], perm, torch.arange(F)[None, :]] = \
                (torch.randint(0, 2, (len(ets), F), generator=g) * 2 - 1).float()
        else:
            W = torch.randn(len(ets), F, F, generator=g) / F ** 0.5
        W = W.to(dtype).cuda()
        rows, cols = {}, {}
        for (s, r, d), c in zip(ets, counts):
            rows[(s, r, d)] = torch.sort(torch.randint(0, min(n_local[s], 200), (c,), generator=g)).values.cuda()
            cols[(s, r, d)] = torch.randint(0, n_local[d], (c,), generator=g).cuda()
        y = rgcn.rgcn_layer_fused_tables(feat, node_id, types, rows, cols, ets, W)
        x = torch.cat([feat[t][node_id[t]] for t in types])
        off = rgcn.type_offsets(n_local, types)
        y_ref = rgcn.rgcn_layer_fused(x, off, rows, cols, ets, W)
        assert y.shape == y_ref.shape == (sum(n_local.values()), F)
        if integer:
            assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
            want = torch.zeros(y.shape, dtype=torch.float64)
            for i, (s, r, d) in enumerate(ets):
                want.index_add_(0, rows[(s, r, d)].cpu() + off[s], x[cols[(s, r, d)] + off[d]].double().cpu() @ W[i].double().cpu())
            assert torch.equal(y.double().cpu(), want)
        else:
            scale = y_ref.float().abs().max().item()
            assert (y.float() - y_ref.float()).abs().max().item() <= 2e-2 * scale


def test_checked_mode_reports_bad_indices(monkeypatch):
    """PYG_HIP_RGCN_CHECK=1: every gather / scatter index is validated on the device (unchecked, a bad index
    is an out-of-bounds DMA read or a packed atomic into foreign memory); index tensors on another device are refused."""
    import ctypes
    import os.path as osp
    from pyg_lib_amd import _capi
    L = _capi.lib()
    x = torch.randn(100, 128, device='cuda').bfloat16()
    w = torch.randn(1, 128, 128, device='cuda').bfloat16()
    out = torch.zeros(50, 128, device='cuda').bfloat16()
    g = torch.randint(0, 100, (300,), device='cuda')
    s = torch.sort(torch.randint(0, 50, (300,), device='cuda')).values

    class Rel(ctypes.Structure):
        _fields_ = [('gather_index', ctypes.c_void_p), ('scatter_index', ctypes.c_void_p), ('num_edges', ctypes.c_int64),
                    ('gather_offset', ctypes.c_int64), ('scatter_offset', ctypes.c_int64), ('weight', ctypes.c_void_p),
                    ('x', ctypes.c_void_p), ('gather_map', ctypes.c_void_p), ('x_rows', ctypes.c_int64),
                    ('gather_map_len', ctypes.c_int64), ('scatter_rows', ctypes.c_int64)]

    L.pyg_hip_rgcn_fused_workspace_size.restype = ctypes.c_size_t
    L.pyg_hip_rgcn_fused_workspace_size.argtypes = [ctypes.c_int64, ctypes.c_int64]
    L.pyg_hip_rgcn_fused.restype = ctypes.c_int
    L.pyg_hip_rgcn_fused.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    ws_bytes = L.pyg_hip_rgcn_fused_workspace_size(1, 300)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream

    def call(gi, si, checked):
        rel = Rel(gi.data_ptr(), si.data_ptr(), gi.numel(), 0, 0, w.data_ptr(), None, None, 0, 0)
        return L.pyg_hip_rgcn_fused(3, x.data_ptr(), 100, ctypes.byref(rel), 1, out.data_ptr(), 50, 128, 128, checked,
                                    ws.data_ptr(), ws_bytes, stream)

    assert call(g, s, 1) == 0
    bad_g = g.clone()
    bad_g[17] = 100
    assert call(bad_g, s, 1) != 0 and b'gather index out of range' in L.pyg_hip_last_error()
    bad_s = s.clone()
    bad_s[-1] = 50
    assert call(g, bad_s, 1) != 0 and b'scatter index out of range' in L.pyg_hip_last_error()
    with pytest.raises(RuntimeError, match='must live on the device'):
        torch.ops.pyg.rgcn_fused(x, [g.cpu()], [s], [0], [0], w, out)


def test_default_mode_validates_indices_without_synchronising():
    """VERDICT r4 weak 4 / ADVICE r3: by default (no PYG_HIP_RGCN_CHECK in the environment) a stale node id must not be an
    out-of-bounds read.  The kernel validates every index, redirects offenders to row 0 and leaves a code in a pinned
    word: `pending_index_error()` returns it once the stream is synchronised, and a later fused call raises for it."""
    import os
    from pyg_lib_amd import rgcn
    if os.environ.get('PYG_HIP_RGCN_CHECK') not in (None, ''):
        pytest.skip('PYG_HIP_RGCN_CHECK is set: not the default mode')
    n_table = 1000
    feat = {'a': torch.randn(n_table, 128, device='cuda').bfloat16()}
    w = torch.eye(128, device='cuda').bfloat16().unsqueeze(0).contiguous()
    node_id = {'a': torch.arange(0, 60, device='cuda')}
    et = ('a', 'r', 'a')
    rows = {et: torch.sort(torch.randint(0, 60, (500,), device='cuda')).values}
    cols = {et: torch.randint(0, 60, (500,), device='cuda')}
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    good = rgcn.rgcn_layer_fused_tables(feat, node_id, ['a'], rows, cols, [et], w)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    stale = {'a': node_id['a'].clone()}
    stale['a'][7] = 10 ** 12                              # a node id far outside the table: would fault unchecked
    y = rgcn.rgcn_layer_fused_tables(feat, stale, ['a'], rows, cols, [et], w)   # does not raise, does not fault
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    assert rgcn.pending_index_error() == 1                # gather index, reported once ...
    assert rgcn.pending_index_error() == 0                # ... and cleared
    y = rgcn.rgcn_layer_fused_tables(feat, stale, ['a'], rows, cols, [et], w)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='earlier call'):   # unpolled: the next call reports it
        rgcn.rgcn_layer_fused_tables(feat, node_id, ['a'], rows, cols, [et], w)
    again = rgcn.rgcn_layer_fused_tables(feat, node_id, ['a'], rows, cols, [et], w)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    assert torch.equal(again.view(torch.int16), good.view(torch.int16)) or \
        (again.float() - good.float()).abs().max() <= 2e-2 * good.float().abs().max()


def _hetero_case(g, dtype, integer):
    types = ['a', 'b', 'c']
    n_global = {'a': 5000, 'b': 300, 'c': 20000}
    n_local = {'a': 700, 'b': 64, 'c': 1500}
    ets = [('a', 'r0', 'a'), ('a', 'r1', 'b'), ('b', 'r2', 'a'), ('c', 'r3', 'a'), ('a', 'r4', 'c'), ('c', 'r5', 'c')]
    counts = [4000, 33, 1000, 0, 129, 9000]
    F = 128
    feat = {t: (torch.randint(-2, 3, (n_global[t], F), generator=g).float() if integer
                else torch.randn(n_global[t], F, generator=g)).to(dtype).cuda() for t in types}
    node_id = {t: torch.randperm(n_global[t], generator=g)[:n_local[t]].cuda() for t in types}
    if integer:
        perm = torch.stack([torch.randperm(F, generator=g) for _ in ets])
        W = torch.zeros(len(ets), F, F)
        W[torch.arange(len(ets))[:, None], perm, torch.arange(F)[None, :]] = \
            (torch.randint(0, 2, (len(ets), F), generator=g) * 2 - 1).float()
    else:
        W = torch.randn(len(ets), F, F, generator=g) / F ** 0.5
    rows, cols = {}, {}
    for (s, r, d), c in zip(ets, counts):
        rows[(s, r, d)] = torch.sort(torch.randint(0, min(n_local[s], 200), (c,), generator=g)).values.cuda()
        cols[(s, r, d)] = torch.randint(0, n_local[d], (c,), generator=g).cuda()
    return types, n_local, ets, feat, node_id, W.to(dtype).cuda(), rows, cols


def _float64_grads(x, W, go, rows, cols, ets, off):
    """dX, dW of out[row] += x[col] @ W_r in float64 (the formulas autograd derives for the reference's chain)."""
    xd, Wd, god = x.double().cpu(), W.double().cpu(), go.double().cpu()
    gx = torch.zeros_like(xd)
    gw = torch.zeros_like(Wd)
    for i, (s, r, d) in enumerate(ets):
        ri, ci = rows[(s, r, d)].cpu() + off[s], cols[(s, r, d)].cpu() + off[d]
        gx.index_add_(0, ci, god[ri] @ Wd[i].t())
        gw[i] = xd[ci].t() @ god[ri]
    return gx, gw


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_fused_layer_is_differentiable_and_matches_the_chain(dtype):
    """VERDICT r3 Missing 4: `rgcn_layer_fused` under autograd keeps the ONE-launch forward (no fallback) and returns the
    gradients of the reference's chain gather_coo -> segment_matmul -> scatter_sum: dX through the same fused kernel with
    swapped index roles and W^T, dW through the weight-gradient kernel on the gathered rows."""
    from pyg_lib_amd import ops, rgcn
    g = torch.Generator().manual_seed(31)
    types, n_local, ets, feat, node_id, W, rows, cols = _hetero_case(g, dtype, integer=False)
    off = rgcn.type_offsets(n_local, types)
    x = torch.cat([feat[t][node_id[t]] for t in types])
    go = torch.randn(x.size(0), 128, generator=g).to(dtype).cuda()
    xf, wf = x.clone().requires_grad_(), W.clone().requires_grad_()
    before = ops.matmul_dw_counters()
    y = rgcn.rgcn_layer_fused(xf, off, rows, cols, ets, wf)
    assert y.grad_fn is not None and type(y.grad_fn).__name__.startswith('_RGCNFused')   # the fused forward, not the chain
    gx, gw = torch.autograd.grad(y, [xf, wf], go)
    assert sum(ops.matmul_dw_counters()) == sum(before) + 1                               # dW: one kernel launch
    # the chain's gradients (same op-level rounding points) and the float64 truth
    xc, wc = x.clone().requires_grad_(), W.clone().requires_grad_()
    yc = rgcn.rgcn_layer(xc, off, rows, cols, ets, wc)
    gxc, gwc = torch.autograd.grad(yc, [xc, wc], go)
    want_x, want_w = _float64_grads(x, W, go, rows, cols, ets, off)
    with torch.no_grad():
        y0 = rgcn.rgcn_layer_fused(x, off, rows, cols, ets, W)
    assert torch.equal(y.detach().view(torch.int16), y0.view(torch.int16)) or \
        (y.float() - y0.float()).abs().max() <= 2e-2 * y0.float().abs().max()   # (atomic order may differ between runs)
    eps = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    for got, chain, want in ((gx, gxc, want_x), (gw, gwc, want_w)):
        scale = want.abs().max().item()
        assert scale > 0.5
        err = (got.double().cpu() - want).abs().max().item()
        err_chain = (chain.double().cpu() - want).abs().max().item()
        assert err <= 4 * eps * scale, (err, scale)
        assert err_chain <= 4 * eps * scale, (err_chain, scale)   # the three-op chain's own backward (unsorted gather)
        assert err <= 2 * err_chain + eps * scale, (err, err_chain)   # as accurate as the chain's own backward


def test_fused_layer_gradients_are_exact_on_integer_data():
    """Small-integer features, signed-permutation weights, small-integer upstream gradients: every partial sum of dX and
    dW is an exact bf16 integer, so both must equal the float64 formulas bit for bit."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(32)
    types, n_local, ets, feat, node_id, W, rows, cols = _hetero_case(g, torch.bfloat16, integer=True)
    off = rgcn.type_offsets(n_local, types)
    x = torch.cat([feat[t][node_id[t]] for t in types])
    go = torch.zeros(x.size(0), 128)
    hot = torch.randint(0, x.size(0), (60,), generator=g)       # a sparse upstream gradient keeps the sums below 256
    go[hot] = torch.randint(-1, 2, (60, 128), generator=g).float()
    go = go.bfloat16().cuda()
    xf, wf = x.clone().requires_grad_(), W.clone().requires_grad_()
    y = rgcn.rgcn_layer_fused(xf, off, rows, cols, ets, wf)
    gx, gw = torch.autograd.grad(y, [xf, wf], go)
    want_x, want_w = _float64_grads(x, W, go, rows, cols, ets, off)
    assert want_x.abs().max() < 256 and want_w.abs().max() < 256
    assert torch.equal(gx.double().cpu(), want_x)
    assert torch.equal(gw.double().cpu(), want_w)


@pytest.mark.parametrize('feat_grad', [False, True])
def test_fused_tables_layer_is_differentiable(feat_grad):
    """`rgcn_layer_fused_tables` under autograd: weight gradient always, table gradients (index_add of the per-batch dX)
    for the tables that ask -- against autograd through the materialised chain."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(33)
    types, n_local, ets, feat, node_id, W, rows, cols = _hetero_case(g, torch.bfloat16, integer=False)
    go = torch.randn(sum(n_local.values()), 128, generator=g).bfloat16().cuda()
    f1 = {t: feat[t].clone().requires_grad_(feat_grad and t != 'b') for t in types}
    w1 = W.clone().requires_grad_()
    y = rgcn.rgcn_layer_fused_tables(f1, node_id, types, rows, cols, ets, w1)
    assert type(y.grad_fn).__name__.startswith('_RGCNFusedTables')
    wanted = [w1] + [f1[t] for t in types if f1[t].requires_grad]
    got = torch.autograd.grad(y, wanted, go)
    f2 = {t: feat[t].clone().requires_grad_(feat_grad and t != 'b') for t in types}
    w2 = W.clone().requires_grad_()
    x2 = torch.cat([f2[t][node_id[t]] for t in types])
    y2 = rgcn.rgcn_layer(x2, rgcn.type_offsets(n_local, types), rows, cols, ets, w2)
    ref = torch.autograd.grad(y2, [w2] + [f2[t] for t in types if f2[t].requires_grad], go)
    assert len(got) == len(ref) == (3 if feat_grad else 1)
    for a, b in zip(got, ref):
        scale = b.float().abs().max().item()
        assert a.shape == b.shape and scale > 0.1
        assert (a.float() - b.float()).abs().max().item() <= 3e-2 * scale


