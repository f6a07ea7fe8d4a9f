"""The atomic-free fused R-GCN layer (PYG_HIP_RGCN_GROUPED, csrc/hip/rgcn_grouped.h): `grouped=True` promises scatter
indices that are nondecreasing per relation -- what the samplers emit -- and the layer is computed owner-computes: no
atomics, no zero fill, every row written once.  Checked here: exact results on integer data (rows of 0 ... 70 edges, rows
of exactly 16 / 32 edges, several node types at offsets that are no multiples of 32, empty relations, more relations than
the kernel argument holds, a feature table beyond 4 GiB), float data against a float64 restatement on a sampled
MAG-shaped neighbourhood, bit-reproducibility, the device-side verification of the promise, autograd, deterministic mode."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def signed_permutations(R, F, g):
    perm = torch.stack([torch.randperm(F, generator=g) for _ in range(R)])
    W = torch.zeros(R, F, F)
    W[torch.arange(R)[:, None], perm, torch.arange(F)[None, :]] = (torch.randint(0, 2, (R, F), generator=g) * 2 - 1).float()
    return W


def column_selectors(R, K, M, g):
    # one +-1 per column: out[:, n] = +- x[:, k(n)], exact whatever the sizes
    W = torch.zeros(R, K, M)
    k = torch.randint(0, K, (R, M), generator=g)
    W[torch.arange(R)[:, None], k, torch.arange(M)[None, :]] = (torch.randint(0, 2, (R, M), generator=g) * 2 - 1).float()
    return W


def exact_want(n_out, F, ets, rows, cols, x, W, soff, goff):
    want = torch.zeros(n_out, F, dtype=torch.float64)
    for i, et in enumerate(ets):
        want.index_add_(0, rows[et].cpu() + soff[i], x[cols[et].cpu() + goff[i]].double() @ W[i].double())
    return want


def test_grouped_integer_valued_inputs_are_exact():
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(21)
    F = 128
    types = ['a', 'b', 'c']
    n = {'a': 301, 'b': 77, 'c': 1000}           # type offsets 0, 301, 378: blocks of 32 rows straddle the types
    ets = [('a', 'r0', 'a'), ('b', 'r1', 'a'), ('a', 'r2', 'b'), ('c', 'r3', 'c'), ('c', 'r4', 'a'), ('a', 'r5', 'c'), ('b', 'r6', 'b')]
    #        rows of ~17 edges   empty          17 edges total     ~70 per row        exactly 16 / 32 / 48    one edge       long tail
    counts = [1000, 0, 17, 4096 + 33, None, 1, 129]
    x = {t: torch.randint(-1, 2, (n[t], F), generator=g).float() for t in types}
    W = signed_permutations(len(ets), F, g)
    rows, cols = {}, {}
    for et, c in zip(ets, counts):
        s, _, d = et
        if c is None:   # rows with exactly 16, 32 and 48 edges next to each other, then rows of 1, then the array's end
            r = torch.cat([torch.full((16,), 3), torch.full((32,), 4), torch.full((48,), 5), torch.arange(6, 40), torch.full((16,), n[s] - 1)])
        else:
            r = torch.sort(torch.randint(0, min(60, n[s]), (c,), generator=g)).values
        rows[et] = r.cuda()
        cols[et] = torch.randint(0, n[d], (r.numel(),), generator=g).cuda()
    off = rgcn.type_offsets(n, types)
    xc = torch.cat([x[t] for t in types])
    soff = [off[s] for s, _, _ in ets]
    goff = [off[d] for _, _, d in ets]
    want = exact_want(off['__total__'], F, ets, rows, cols, xc, W, soff, goff)
    assert want.abs().max() <= 256   # every feature sum and every result exactly representable in bf16
    for dtype in (torch.bfloat16, torch.float16):
        y = rgcn.rgcn_layer_fused(xc.to(dtype).cuda(), off, rows, cols, ets, W.to(dtype).cuda(), grouped=True)
        assert y.dtype == dtype and torch.equal(y.double().cpu(), want)
        # ... and the atomic kernel agrees bit for bit on such data
        y2 = rgcn.rgcn_layer_fused(xc.to(dtype).cuda(), off, rows, cols, ets, W.to(dtype).cuda())
        assert torch.equal(y, y2)
    # from the global tables through node ids
    n_glob = {'a': 5000, 'b': 900, 'c': 20000}
    nid = {t: torch.randperm(n_glob[t], generator=g)[:n[t]] for t in types}
    tab = {t: torch.randint(-3, 4, (n_glob[t], F), generator=g).float() for t in types}
    for t in types:
        tab[t][nid[t]] = x[t]
    yt = rgcn.rgcn_layer_fused_tables({t: tab[t].bfloat16().cuda() for t in types}, {t: nid[t].cuda() for t in types}, types,
                                      rows, cols, ets, W.bfloat16().cuda(), grouped=True)
    assert torch.equal(yt.double().cpu(), want)
    assert rgcn.pending_index_error() == 0


def test_grouped_more_relations_than_the_kernel_argument_holds():
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(13)
    n, F, R = 900, 128, 30
    x = torch.randint(-1, 2, (n, F), generator=g).float()
    counts = [int(c) for c in torch.randint(0, 300, (R,), generator=g)]
    counts[3] = 0
    counts[7] = 5
    counts[29] = 700
    ets = [('a', f'r{i}', 'a') for i in range(R)]
    W = signed_permutations(R, F, g)
    rows, cols = {}, {}
    for i, (et, c) in enumerate(zip(ets, counts)):
        rows[et] = torch.sort(torch.randint(0, 50 + 25 * i, (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, n, (c,), generator=g).cuda()
    off = rgcn.type_offsets({'a': n}, ['a'])
    y = rgcn.rgcn_layer_fused(x.bfloat16().cuda(), off, rows, cols, ets, W.bfloat16().cuda(), grouped=True)
    want = exact_want(n, F, ets, rows, cols, x, W, [0] * R, [0] * R)
    assert want.abs().max() <= 256
    assert torch.equal(y.double().cpu(), want)


@pytest.mark.parametrize('R', [70, 200, 512, 513])
def test_grouped_many_relations(R):
    """The relations with edges into a block are found 64 at a time (one ballot per 64 relations); 512 is the most the
    kernel keeps row ranges for -- beyond that the layer function takes the atomic kernel.  Integer data, exact."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(1000 + R)
    n, F = 500, 128
    x = torch.randint(-1, 2, (n, F), generator=g).float()
    ets = [('a', f'r{i}', 'a') for i in range(R)]
    W = signed_permutations(R, F, g)
    rows, cols = {}, {}
    for i, et in enumerate(ets):
        c = int(torch.randint(0, 40, (1,), generator=g)) if i % 7 else 0      # every seventh relation is empty
        lo = int(torch.randint(0, n - 40, (1,), generator=g))                 # each relation touches its own stretch of rows
        rows[et] = torch.sort(torch.randint(lo, lo + 40, (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, n, (c,), generator=g).cuda()
    off = rgcn.type_offsets({'a': n}, ['a'])
    y = rgcn.rgcn_layer_fused(x.bfloat16().cuda(), off, rows, cols, ets, W.bfloat16().cuda(), grouped=True)
    want = exact_want(n, F, ets, rows, cols, x, W, [0] * R, [0] * R)
    assert want.abs().max() <= 256
    assert torch.equal(y.double().cpu(), want)


def test_grouped_random_configurations():
    """Differential fuzz of the atomic-free kernels: random numbers of node types and relations, offsets that are no
    multiples of 16, empty relations, rows of 0 ... 100 edges, edges at the very end of their arrays, every (dtype, K, M)
    the kernels take, x or tables -- integer data, exact against float64."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(20260927)
    shapes = [(torch.bfloat16, 128, 128), (torch.float16, 128, 128), (torch.bfloat16, 256, 256), (torch.float16, 128, 256),
              (torch.bfloat16, 256, 128), (torch.float32, 128, 128), (torch.bfloat16, 64, 64), (torch.float16, 128, 64),
              (torch.bfloat16, 64, 128), (torch.float16, 40, 216), (torch.bfloat16, 192, 24), (torch.float16, 256, 16)]
    for case in range(72):
        dtype, K, M = shapes[case % len(shapes)]
        T = int(torch.randint(1, 4, (1,), generator=g))
        types = [f't{i}' for i in range(T)]
        n = {t: int(torch.randint(1, 300, (1,), generator=g)) for t in types}
        R = int(torch.randint(1, 9, (1,), generator=g))
        ets, rows, cols = [], {}, {}
        for i in range(R):
            s = types[int(torch.randint(0, T, (1,), generator=g))]
            d = types[int(torch.randint(0, T, (1,), generator=g))]
            et = (s, f'r{i}', d)
            ets.append(et)
            mode = int(torch.randint(0, 4, (1,), generator=g))
            c = [0, int(torch.randint(1, 20, (1,), generator=g)), int(torch.randint(20, 400, (1,), generator=g)),
                 int(torch.randint(400, 3000, (1,), generator=g))][mode]
            hi = max(1, int(torch.randint(1, n[s] + 1, (1,), generator=g)))
            r = torch.sort(torch.randint(0, hi, (c,), generator=g)).values
            if c and mode == 3:
                r[-1] = n[s] - 1            # an edge into the type's last row
                r = torch.sort(r).values
            rows[et] = r.cuda()
            cols[et] = torch.randint(0, n[d], (c,), generator=g).cuda()
        off = rgcn.type_offsets(n, types)
        x = torch.cat([torch.randint(-1, 2, (n[t], K), generator=g).float() for t in types])
        W = column_selectors(R, K, M, g)
        soff = [off[s] for s, _, _ in ets]
        goff = [off[d] for _, _, d in ets]
        want = exact_want(off['__total__'], M, ets, rows, cols, x, W, soff, goff)
        if want.abs().max() > 256:
            continue
        y = rgcn.rgcn_layer_fused(x.to(dtype).cuda(), off, rows, cols, ets, W.to(dtype).cuda(), grouped=True)
        assert y.shape == (off['__total__'], M) and torch.equal(y.double().cpu(), want), (case, dtype, K, M, n, [(e, rows[e].numel()) for e in ets])
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0


def test_grouped_without_edges_writes_zeros():
    from pyg_lib_amd import rgcn
    ets = [('a', 'x', 'a'), ('a', 'y', 'a')]
    e = torch.zeros(0, dtype=torch.long, device='cuda')
    x = torch.randn(70, 128, device='cuda').bfloat16()
    w = torch.randn(2, 128, 128, device='cuda').bfloat16()
    off = rgcn.type_offsets({'a': 70}, ['a'])
    for _ in range(3):   # (the output is NOT zero-filled by the caller in this mode: stale allocator blocks must not show)
        torch.full((70, 128), 7.0, device='cuda').bfloat16()
        y = rgcn.rgcn_layer_fused(x, off, {et: e for et in ets}, {et: e for et in ets}, ets, w, grouped=True)
        assert y.shape == (70, 128) and not y.any()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_grouped_on_a_sampled_mag_neighbourhood(dtype):
    """The sampler's output IS grouped (row nondecreasing per edge type): verified by the device-side check of the call,
    result against a float64 restatement with the kernel's two roundings (feature sum per relation, result), against the
    atomic kernel and the three-op chain, untouched rows exactly zero, the same bits on every run."""
    from pyg_lib_amd import sampler, rgcn
    from tests.test_rgcn_gpu import MAG_TYPES, MAG_ETS, build_graph
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(3)
    sizes = {'paper': 40_000, 'author': 60_000, 'institution': 900, 'field_of_study': 4_000}
    rp, cl = build_graph(rng, sizes, MAG_ETS, 12)
    seeds = {'paper': rng.permutation(sizes['paper'])[:1024].astype(np.int64)}
    fan = {e: [15, 10] for e in MAG_ETS}
    F = 128
    torch.manual_seed(9)
    out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in rp.items()}, {e: dev(v) for e, v in cl.items()},
                                         {k: dev(v) for k, v in seeds.items()}, fan)
    row_d, col_d, node_d = out[0], out[1], out[2]
    for e in MAG_ETS:
        assert bool((row_d[e][1:] >= row_d[e][:-1]).all()), e   # the promise holds for sampler output
    g = torch.Generator(device='cuda').manual_seed(4)
    feat = {t: torch.randn(sizes[t], F, device='cuda', generator=g).to(dtype) for t in MAG_TYPES}
    W = (torch.randn(len(MAG_ETS), F, F, device='cuda', generator=g) / F ** 0.5).to(dtype)
    off = rgcn.type_offsets({t: node_d[t].numel() for t in MAG_TYPES}, MAG_TYPES)
    x = torch.cat([feat[t][node_d[t]] for t in MAG_TYPES])
    y = rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, grouped=True)
    yt = rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, W, grouped=True)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    assert torch.equal(y, yt)           # same sums in the same order, wherever the rows come from
    want = torch.zeros(off['__total__'], F, dtype=torch.float64, device='cuda')
    touched = torch.zeros(off['__total__'], dtype=torch.bool, device='cuda')
    for i, (s, r, d) in enumerate(MAG_ETS):
        agg = torch.zeros(off['__total__'], F, dtype=torch.float64, device='cuda')
        agg.index_add_(0, row_d[(s, r, d)] + off[s], x[col_d[(s, r, d)] + off[d]].double())
        want += agg.to(dtype).double() @ W[i].double()           # the feature sum is rounded to T once
        touched[row_d[(s, r, d)] + off[s]] = True
    scale = want.abs().max().item()
    assert scale > 1.0
    # one rounding of the result (half an ulp: 2^-8 / 2^-11 relative) on top of the restated feature-sum rounding (an fp32
    # sum next to a rounding boundary may land on the other side of it than the float64 sum: one ulp of one feature sum)
    assert (y.double() - want).abs().max().item() <= (8e-3 if dtype == torch.bfloat16 else 1.5e-3) * scale
    assert not y[~touched].any()
    ya = rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, grouped=False)     # atomic kernel: messages rounded, runs added
    y3 = rgcn.rgcn_layer(x, off, row_d, col_d, MAG_ETS, W)           # three-op chain
    tol = (3e-2 if dtype == torch.bfloat16 else 4e-3) * scale
    assert (y.double() - ya.double()).abs().max().item() <= tol and (y.double() - y3.double()).abs().max().item() <= tol
    for _ in range(3):
        assert torch.equal(rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, W, grouped=True), y)


@pytest.mark.parametrize('K,M', [(256, 256), (128, 256), (256, 128), (64, 64), (128, 64), (64, 128), (32, 32), (8, 256), (96, 160),
                                 (200, 72), (256, 64), (64, 256), (248, 8), (136, 136)])
def test_grouped_feature_widths_of_64_and_256(K, M):
    """K and M of the atomic-free kernel may be any multiples of 8 up to 256 (256: C4's width): the feature rows are
    walked once per 128-feature slice, W travels through LDS in 128 x 128 chunks; outside {128, 256} lanes / columns / rows
    behind the end are masked (one instance with run-time row sizes).  Exact on integer data (rows of 0 ... 70 edges, several
    node types, through x and through the global tables), a float64 restatement on random data, autograd (the backward
    takes the chain for these shapes), and the ungrouped call still falls back to the three-op chain."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(77 + K + 2 * M)
    types = ['a', 'b']
    n = {'a': 333, 'b': 90}
    ets = [('a', 'r0', 'a'), ('b', 'r1', 'a'), ('a', 'r2', 'b'), ('b', 'r3', 'b')]
    counts = [3000, 40, 4096 + 33, 0]
    x = {t: torch.randint(-1, 2, (n[t], K), generator=g).float() for t in types}
    W = column_selectors(len(ets), K, M, g)
    rows, cols = {}, {}
    for et, c in zip(ets, counts):
        s, _, d = et
        rows[et] = torch.sort(torch.randint(0, min(60, n[s]), (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, n[d], (c,), generator=g).cuda()
    off = rgcn.type_offsets(n, types)
    xc = torch.cat([x[t] for t in types])
    soff = [off[s] for s, _, _ in ets]
    goff = [off[d] for _, _, d in ets]
    want = exact_want(off['__total__'], M, ets, rows, cols, xc, W, soff, goff)
    assert want.abs().max() <= 256
    for dtype in (torch.bfloat16, torch.float16):
        y = rgcn.rgcn_layer_fused(xc.to(dtype).cuda(), off, rows, cols, ets, W.to(dtype).cuda(), grouped=True)
        assert y.shape == (off['__total__'], M) and y.dtype == dtype
        assert torch.equal(y.double().cpu(), want)
    n_glob = {'a': 4000, 'b': 700}
    nid = {t: torch.randperm(n_glob[t], generator=g)[:n[t]] for t in types}
    tab = {t: torch.randint(-3, 4, (n_glob[t], K), generator=g).float() for t in types}
    for t in types:
        tab[t][nid[t]] = x[t]
    yt = rgcn.rgcn_layer_fused_tables({t: tab[t].bfloat16().cuda() for t in types}, {t: nid[t].cuda() for t in types}, types,
                                      rows, cols, ets, W.bfloat16().cuda(), grouped=True)
    assert torch.equal(yt.double().cpu(), want)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    # random data: one rounding of every per-relation feature sum, one of the result
    xr = torch.randn(off['__total__'], K, generator=g).bfloat16().cuda()
    wr = (torch.randn(len(ets), K, M, generator=g) / K ** 0.5).bfloat16().cuda()
    yr = rgcn.rgcn_layer_fused(xr, off, rows, cols, ets, wr, grouped=True)
    ref = torch.zeros(off['__total__'], M, dtype=torch.float64, device='cuda')
    for i, et in enumerate(ets):
        agg = torch.zeros(off['__total__'], K, dtype=torch.float64, device='cuda')
        agg.index_add_(0, rows[et] + soff[i], xr[cols[et] + goff[i]].double())
        ref += agg.bfloat16().double() @ wr[i].double()
    scale = ref.abs().max().item()
    assert scale > 1.0 and (yr.double() - ref).abs().max().item() <= 8e-3 * scale
    assert torch.equal(yr, rgcn.rgcn_layer_fused(xr, off, rows, cols, ets, wr, grouped=True))
    y3 = rgcn.rgcn_layer_fused(xr, off, rows, cols, ets, wr)          # not grouped: the three-op chain for these widths
    # (the chain's scatter_sum adds rows narrower than 64 bytes in the storage type, ~70 bf16 additions per row here in the
    # order the atomics land: measured up to 3.3e-2 of the scale at M = 8 -- one run in ~10 -- against 1e-2 for wide rows)
    assert (yr.double() - y3.double()).abs().max().item() <= (3e-2 if M * 2 >= 64 else 8e-2) * scale
    # autograd
    xg = xr.clone().requires_grad_()
    wg = wr.clone().requires_grad_()
    rgcn.rgcn_layer_fused(xg, off, rows, cols, ets, wg, grouped=True).float().square().sum().backward()
    xg3 = xr.clone().requires_grad_()
    wg3 = wr.clone().requires_grad_()
    rgcn.rgcn_layer(xg3, off, rows, cols, ets, wg3).float().square().sum().backward()
    for a, b in ((xg.grad, xg3.grad), (wg.grad, wg3.grad)):
        assert a.shape == b.shape and (a.float() - b.float()).abs().max().item() <= 6e-2 * b.float().abs().max().item()


@pytest.mark.parametrize('dtype,K,M', [(torch.bfloat16, 128, 128), (torch.bfloat16, 256, 256), (torch.float16, 256, 256),
                                       (torch.float16, 128, 256), (torch.bfloat16, 256, 128), (torch.float32, 128, 128),
                                       (torch.bfloat16, 64, 64), (torch.float16, 192, 40), (torch.bfloat16, 32, 200), (torch.float32, 64, 64)])
def test_grouped_short_rows_take_the_pipeline(dtype, K, M):
    """Rows of at most 16 edges per relation -- what the samplers emit for fan-outs up to 16 -- run through the software
    pipeline of the atomic-free kernel (K, M in {128, 256} and float32 128 x 128; the other widths in the list keep the
    item-at-a-time walk whatever the rows are: built with the pipeline they need 169 - 173 registers, two waves per SIMD,
    and are 1.3 - 1.45 x slower) (one sub-item per 128-feature slice for the 16-bit shapes; float32 with both slices
    and all of W at once, the product on fp32 MFMAs), the other tests' rows of up to 70 edges through the item-at-a-time
    walk.  Enough blocks that every workgroup of the persistent launch owns several items (the pipeline is four deep),
    degrees 0 ... 16 incl. exactly 16, type offsets that are no multiples of 16, an empty relation; exact on integer data
    through x and through the global tables, float data against float64, the same bits on every run."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(1000 + K + 2 * M + (7 if dtype == torch.float32 else 0))
    types = ['a', 'b', 'c']
    n = {'a': 30011, 'b': 9973, 'c': 517}
    ets = [('a', 'r0', 'a'), ('b', 'r1', 'a'), ('a', 'r2', 'b'), ('c', 'r3', 'a'), ('b', 'r4', 'b'), ('a', 'r5', 'c'), ('c', 'r6', 'c')]
    rows, cols = {}, {}
    for i, et in enumerate(ets):
        s, _, d = et
        active = [n[s], n[s] // 2, 2000, 300, 0, n[s], 40][i]                  # rows 0 ... active - 1 may have edges
        deg = torch.randint(0, 17, (active,), generator=g)
        if active:
            deg[torch.randint(0, active, (max(1, active // 8),), generator=g)] = 16
        rows[et] = torch.repeat_interleave(torch.arange(active), deg).cuda()
        cols[et] = torch.randint(0, n[d], (int(deg.sum()),), generator=g).cuda()
    off = rgcn.type_offsets(n, types)
    soff = [off[s] for s, _, _ in ets]
    goff = [off[d] for _, _, d in ets]
    x = {t: torch.randint(-1, 2, (n[t], K), generator=g).float() for t in types}
    xc = torch.cat([x[t] for t in types])
    W = column_selectors(len(ets), K, M, g)
    want = exact_want(off['__total__'], M, ets, rows, cols, xc, W, soff, goff)
    assert want.abs().max() <= 256
    y = rgcn.rgcn_layer_fused(xc.to(dtype).cuda(), off, rows, cols, ets, W.to(dtype).cuda(), grouped=True)
    assert y.shape == (off['__total__'], M) and y.dtype == dtype and torch.equal(y.double().cpu(), want)
    n_glob = {'a': 50000, 'b': 12000, 'c': 600}
    nid = {t: torch.randperm(n_glob[t], generator=g)[:n[t]] for t in types}
    tab = {t: torch.randint(-3, 4, (n_glob[t], K), generator=g).float() for t in types}
    for t in types:
        tab[t][nid[t]] = x[t]
    yt = rgcn.rgcn_layer_fused_tables({t: tab[t].to(dtype).cuda() for t in types}, {t: nid[t].cuda() for t in types}, types,
                                      rows, cols, ets, W.to(dtype).cuda(), grouped=True)
    assert torch.equal(yt.double().cpu(), want)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    # float data
    xr = torch.randn(off['__total__'], K, generator=g).to(dtype).cuda()
    wr = (torch.randn(len(ets), K, M, generator=g) / K ** 0.5).to(dtype).cuda()
    yr = rgcn.rgcn_layer_fused(xr, off, rows, cols, ets, wr, grouped=True)
    ref = torch.zeros(off['__total__'], M, dtype=torch.float64, device='cuda')
    for i, et in enumerate(ets):
        agg = torch.zeros(off['__total__'], K, dtype=torch.float64, device='cuda')
        agg.index_add_(0, rows[et] + soff[i], xr[cols[et] + goff[i]].double())
        ref += (agg if dtype == torch.float32 else agg.to(dtype).double()) @ wr[i].double()   # (16-bit: the feature sum is rounded once)
    scale = ref.abs().max().item()
    tol = {torch.float32: 1e-5, torch.bfloat16: 8e-3, torch.float16: 1.5e-3}[dtype]
    assert scale > 1.0 and (yr.double() - ref).abs().max().item() <= tol * scale
    assert torch.equal(yr, rgcn.rgcn_layer_fused(xr, off, rows, cols, ets, wr, grouped=True))


def test_grouped_float32():
    """float32, F = 128: sums, A tile and product in fp32 (rows of more than 16 edges: plain FMAs; the pipeline of the short
    rows: fp32 MFMAs -- fp32 operands and accumulation either way) -- 1e-5 of the
    scale against a float64 restatement (the tolerance of the reference's own fp32 tests), exact on integer data, rows of
    0 ... 70 edges, several node types, x and tables, the same bits on every run, autograd against the chain."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(404)
    F = 128
    types = ['a', 'b', 'c']
    n = {'a': 301, 'b': 77, 'c': 1000}
    ets = [('a', 'r0', 'a'), ('b', 'r1', 'a'), ('a', 'r2', 'b'), ('c', 'r3', 'c'), ('c', 'r4', 'a'), ('a', 'r5', 'c'), ('b', 'r6', 'b')]
    counts = [1000, 0, 17, 4096 + 33, 300, 1, 129]
    rows, cols = {}, {}
    for et, c in zip(ets, counts):
        s, _, d = et
        rows[et] = torch.sort(torch.randint(0, min(60, n[s]), (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, n[d], (c,), generator=g).cuda()
    off = rgcn.type_offsets(n, types)
    soff = [off[s] for s, _, _ in ets]
    goff = [off[d] for _, _, d in ets]
    # integer data: exact
    xi = torch.randint(-3, 4, (off['__total__'], F), generator=g).float()
    Wi = signed_permutations(len(ets), F, g)
    want = exact_want(off['__total__'], F, ets, rows, cols, xi, Wi, soff, goff)
    y = rgcn.rgcn_layer_fused(xi.cuda(), off, rows, cols, ets, Wi.cuda(), grouped=True)
    assert y.dtype == torch.float32 and torch.equal(y.double().cpu(), want)
    # random data: 1e-5 of the scale
    x = torch.randn(off['__total__'], F, generator=g).cuda()
    W = (torch.randn(len(ets), F, F, generator=g) / F ** 0.5).cuda()
    y = rgcn.rgcn_layer_fused(x, off, rows, cols, ets, W, grouped=True)
    ref = torch.zeros(off['__total__'], F, dtype=torch.float64, device='cuda')
    for i, et in enumerate(ets):
        ref.index_add_(0, rows[et] + soff[i], x[cols[et] + goff[i]].double() @ W[i].double())
    scale = ref.abs().max().item()
    assert scale > 1.0 and (y.double() - ref).abs().max().item() <= 1e-5 * scale
    assert torch.equal(y, rgcn.rgcn_layer_fused(x, off, rows, cols, ets, W, grouped=True))
    y3 = rgcn.rgcn_layer_fused(x, off, rows, cols, ets, W)          # not grouped: the three-op chain for this type
    assert (y3.double() - ref).abs().max().item() <= 1e-5 * scale
    # through the global tables
    n_glob = {'a': 5000, 'b': 900, 'c': 20000}
    nid = {t: torch.randperm(n_glob[t], generator=g)[:n[t]].cuda() for t in types}
    tab = {t: torch.randn(n_glob[t], F, generator=g).cuda() for t in types}
    xt = torch.cat([tab[t][nid[t]] for t in types])
    yt = rgcn.rgcn_layer_fused_tables(tab, nid, types, rows, cols, ets, W, grouped=True)
    assert torch.equal(yt, rgcn.rgcn_layer_fused(xt, off, rows, cols, ets, W, grouped=True))
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    # autograd
    xg = x.clone().requires_grad_()
    wg = W.clone().requires_grad_()
    rgcn.rgcn_layer_fused(xg, off, rows, cols, ets, wg, grouped=True).square().sum().backward()
    xg3 = x.clone().requires_grad_()
    wg3 = W.clone().requires_grad_()
    rgcn.rgcn_layer(xg3, off, rows, cols, ets, wg3).square().sum().backward()
    for a, b in ((xg.grad, xg3.grad), (wg.grad, wg3.grad)):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 1e-4 * b.abs().max().item()


@pytest.mark.parametrize('K,M', [(64, 64), (32, 128), (128, 32), (4, 4), (100, 52), (124, 128), (72, 12)])
def test_grouped_float32_other_widths(K, M):
    """float32 with K, M any multiples of 4 up to 128 (rows of multiples of 16 bytes): the instance with run-time row sizes --
    lanes, W rows and columns behind the ends masked, W's masked rows written as zeros.  Exact on integer data (rows of 0 ... 70
    edges and rows of at most 16, x and tables), 1e-5 against float64 on random data, the same bits on every run."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(900 + K + 3 * M)
    types = ['a', 'b']
    n = {'a': 333, 'b': 90}
    ets = [('a', 'r0', 'a'), ('b', 'r1', 'a'), ('a', 'r2', 'b'), ('b', 'r3', 'b')]
    for counts, hi in (([3000, 40, 4096 + 33, 0], 60), ([300, 40, 400, 7], 10 ** 6)):
        rows, cols = {}, {}
        for et, c in zip(ets, counts):
            s, _, d = et
            rows[et] = torch.sort(torch.randint(0, min(hi, n[s]), (c,), generator=g)).values.cuda()
            cols[et] = torch.randint(0, n[d], (c,), generator=g).cuda()
        off = rgcn.type_offsets(n, types)
        soff = [off[s] for s, _, _ in ets]
        goff = [off[d] for _, _, d in ets]
        x = {t: torch.randint(-3, 4, (n[t], K), generator=g).float() for t in types}
        xc = torch.cat([x[t] for t in types])
        W = column_selectors(len(ets), K, M, g)
        want = exact_want(off['__total__'], M, ets, rows, cols, xc, W, soff, goff)
        y = rgcn.rgcn_layer_fused(xc.cuda(), off, rows, cols, ets, W.cuda(), grouped=True)
        assert rgcn.last_layer_path() == 'grouped'
        assert y.shape == (off['__total__'], M) and y.dtype == torch.float32 and torch.equal(y.double().cpu(), want)
        n_glob = {'a': 4000, 'b': 700}
        nid = {t: torch.randperm(n_glob[t], generator=g)[:n[t]] for t in types}
        tab = {t: torch.randint(-3, 4, (n_glob[t], K), generator=g).float() for t in types}
        for t in types:
            tab[t][nid[t]] = x[t]
        yt = rgcn.rgcn_layer_fused_tables({t: tab[t].cuda() for t in types}, {t: nid[t].cuda() for t in types}, types, rows, cols, ets,
                                          W.cuda(), grouped=True)
        assert torch.equal(yt.double().cpu(), want)
        xr = torch.randn(off['__total__'], K, generator=g).cuda()
        wr = (torch.randn(len(ets), K, M, generator=g) / K ** 0.5).cuda()
        yr = rgcn.rgcn_layer_fused(xr, off, rows, cols, ets, wr, grouped=True)
        ref = torch.zeros(off['__total__'], M, dtype=torch.float64, device='cuda')
        for i, et in enumerate(ets):
            ref.index_add_(0, rows[et] + soff[i], xr[cols[et] + goff[i]].double() @ wr[i].double())
        scale = ref.abs().max().item()
        assert scale > 0.5 and (yr.double() - ref).abs().max().item() <= 1e-5 * scale
        assert torch.equal(yr, rgcn.rgcn_layer_fused(xr, off, rows, cols, ets, wr, grouped=True))
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0


@pytest.mark.parametrize('short_rows', [True, False])
def test_grouped_float32_special_values(short_rows):
    """float32 through both products of the atomic-free kernel (rows of at most 16 edges: the pipeline's fp32 MFMAs; longer
    rows: plain FMAs): denormal features stay denormal (their sums and +-1 weights are exact), Inf and NaN propagate to the
    rows that gather them and to no others -- the same as the float64 restatement rounded to float32."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(77 if short_rows else 78)
    F = 128
    n = 20000 if short_rows else 600
    ets = [('a', 'r0', 'a'), ('a', 'r1', 'a')]
    rows, cols = {}, {}
    for et in ets:
        deg = torch.randint(0, 17 if short_rows else 60, (n // 2,), generator=g)
        rows[et] = torch.repeat_interleave(torch.arange(n // 2), deg).cuda()
        cols[et] = torch.randint(0, n, (int(deg.sum()),), generator=g).cuda()
    off = rgcn.type_offsets({'a': n}, ['a'])
    tiny = 2.0 ** -149
    x = (torch.randint(-40, 41, (n, F), generator=g).double() * tiny)
    x[3, 5], x[7, 100], x[11, 64] = float('inf'), float('nan'), float('-inf')
    W = signed_permutations(len(ets), F, g)
    want = torch.zeros(n, F, dtype=torch.float64)
    for i, et in enumerate(ets):
        want.index_add_(0, rows[et].cpu(), x[cols[et].cpu()] @ W[i].double())
    y = rgcn.rgcn_layer_fused(x.float().cuda(), off, rows, cols, ets, W.cuda(), grouped=True).cpu()
    assert rgcn.last_layer_path() == 'grouped'
    wf = want.float()
    assert torch.equal(torch.isnan(y), torch.isnan(wf))
    assert int(torch.isnan(wf).sum()) > 0 and int(torch.isinf(wf).sum()) > 0
    ok = ~torch.isnan(wf)
    assert torch.equal(y[ok], wf[ok])
    assert float(wf[ok & ~torch.isinf(wf)].abs().max()) < 1e-40 and float(wf[ok & ~torch.isinf(wf)].abs().max()) > 0   # all denormal


def test_sampler_rows_take_the_atomic_free_kernel_by_default():
    """`grouped=None` (the default): rows that ARE outputs of this package's samplers (csc=False) are nondecreasing by
    construction, so the layer runs the atomic-free kernel without a flag; a copy of them (or any hand-made edge list) is
    not known to be grouped and takes the atomic kernel."""
    from pyg_lib_amd import sampler, rgcn, diagnostics
    from tests.test_rgcn_gpu import MAG_TYPES, MAG_ETS, build_graph
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(8)
    sizes = {'paper': 4000, 'author': 6000, 'institution': 90, 'field_of_study': 400}
    rp, cl = build_graph(rng, sizes, MAG_ETS, 12)
    seeds = {'paper': dev(rng.permutation(sizes['paper'])[:64].astype(np.int64))}
    out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in rp.items()}, {e: dev(v) for e, v in cl.items()}, seeds,
                                         {e: [5, 4] for e in MAG_ETS})
    row_d, col_d, node_d = out[0], out[1], out[2]
    assert all(sampler.rows_are_grouped(row_d[e]) for e in MAG_ETS) and not sampler.rows_are_grouped(col_d[MAG_ETS[0]])
    g = torch.Generator(device='cuda').manual_seed(4)
    feat = {t: torch.randn(sizes[t], 128, device='cuda', generator=g).bfloat16() for t in MAG_TYPES}
    W = (torch.randn(len(MAG_ETS), 128, 128, device='cuda', generator=g) / 11).bfloat16()
    ya = rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, W, grouped=False)
    marker = diagnostics.last_accumulate_info()
    assert 'pyg_hip_rgcn_fused' in marker
    torch.zeros(8, device='cuda').index_add_(0, torch.zeros(3, dtype=torch.long, device='cuda'), torch.ones(3, device='cuda'))
    from pyg_lib_amd import ops
    ops.scatter_sum(torch.ones(4, 2, device='cuda'), torch.tensor([[0, 1], [1, 0], [0, 0], [1, 1]], device='cuda'), 0, None, 2)
    marker = diagnostics.last_accumulate_info()        # (an atomic scatter in between: the marker is no longer the layer's)
    assert 'pyg_hip_rgcn_fused' not in marker
    y = rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, W)          # default: atomic-free
    assert diagnostics.last_accumulate_info() == marker
    assert torch.equal(y, rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, W, grouped=True))
    scale = ya.float().abs().max().item()
    assert (y.float() - ya.float()).abs().max().item() <= 3e-2 * scale
    copies = {e: row_d[e].clone() for e in MAG_ETS}
    rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, copies, col_d, MAG_ETS, W)             # copies: the atomic kernel
    assert 'pyg_hip_rgcn_fused' in diagnostics.last_accumulate_info()
    # csc=True: `row` holds the sampled neighbours (not grouped, not remembered), `col` the expanded nodes (remembered)
    n = 3000
    deg = torch.randint(0, 9, (n,), generator=torch.Generator().manual_seed(2))
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), deg.cumsum(0)]).cuda()
    idx = torch.randint(0, n, (int(deg.sum()),), generator=torch.Generator().manual_seed(3)).cuda()
    sd = torch.arange(0, 200, 5).cuda()
    assert sampler.rows_are_grouped(sampler.neighbor_sample(ptr, idx, sd, [4, 3])[0])
    oc = sampler.neighbor_sample(ptr, idx, sd, [4, 3], csc=True)
    assert not sampler.rows_are_grouped(oc[0]) and sampler.rows_are_grouped(oc[1]) and bool((oc[1][1:] >= oc[1][:-1]).all())
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0


def test_grouped_promise_is_verified_on_the_device():
    """Not grouped -> error 3: reported without a synchronisation by default (pending_index_error / the next call), in
    the call itself with PYG_HIP_RGCN_CHECKED; an out-of-range scatter index is 2, a gather index 1; nothing is read or
    written out of bounds (the arrays sit at the end of their allocations here)."""
    from pyg_lib_amd import rgcn, _capi
    g = torch.Generator().manual_seed(5)
    n, F = 500, 128
    ets = [('a', 'r', 'a')]
    x = torch.randn(n, F, generator=g).bfloat16().cuda()
    w = torch.randn(1, F, F, generator=g).bfloat16().cuda()
    off = rgcn.type_offsets({'a': n}, ['a'])
    rows = torch.sort(torch.randint(0, n, (3000,), generator=g)).values
    cols = torch.randint(0, n, (3000,), generator=g)
    assert rgcn.pending_index_error() == 0
    rgcn.rgcn_layer_fused(x, off, {ets[0]: rows.cuda()}, {ets[0]: cols.cuda()}, ets, w, grouped=True)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    shuffled = rows[torch.randperm(3000, generator=g)]
    rgcn.rgcn_layer_fused(x, off, {ets[0]: shuffled.cuda()}, {ets[0]: cols.cuda()}, ets, w, grouped=True)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 3
    bad_rows = rows.clone()
    bad_rows[-1] = n + 5
    rgcn.rgcn_layer_fused(x, off, {ets[0]: bad_rows.cuda()}, {ets[0]: cols.cuda()}, ets, w, grouped=True)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 2
    bad_cols = cols.clone()
    bad_cols[17] = -3
    y = rgcn.rgcn_layer_fused(x, off, {ets[0]: rows.cuda()}, {ets[0]: bad_cols.cuda()}, ets, w, grouped=True)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 1 and bool(torch.isfinite(y.float()).all())
    # unpolled: the next call raises
    rgcn.rgcn_layer_fused(x, off, {ets[0]: shuffled.cuda()}, {ets[0]: cols.cuda()}, ets, w, grouped=True)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='not grouped'):
        rgcn.rgcn_layer_fused(x, off, {ets[0]: rows.cuda()}, {ets[0]: cols.cuda()}, ets, w, grouped=True)
    assert rgcn.pending_index_error() == 0

    # the C entry point with PYG_HIP_RGCN_CHECKED | PYG_HIP_RGCN_GROUPED fails in the call that has them
    L = _capi.lib()

    class Rel(ctypes.Structure):
        _fields_ = [('gather_index', ctypes.c_void_p), ('scatter_index', ctypes.c_void_p), ('num_edges', ctypes.c_int64),
                    ('gather_offset', ctypes.c_int64), ('scatter_offset', ctypes.c_int64), ('weight', ctypes.c_void_p),
                    ('x', ctypes.c_void_p), ('gather_map', ctypes.c_void_p), ('x_rows', ctypes.c_int64), ('gather_map_len', ctypes.c_int64), ('scatter_rows', ctypes.c_int64)]

    L.pyg_hip_rgcn_grouped_workspace_size.restype = ctypes.c_size_t
    L.pyg_hip_rgcn_grouped_workspace_size.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    L.pyg_hip_rgcn_fused.restype = ctypes.c_int
    L.pyg_hip_rgcn_fused.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    out = torch.empty(n, F, dtype=torch.bfloat16, device='cuda')

    def call(s, gidx, flags):
        sd, gd = s.cuda(), gidx.cuda()
        rel = Rel(gd.data_ptr(), sd.data_ptr(), sd.numel(), 0, 0, w.data_ptr(), 0, 0, 0, 0)
        need = L.pyg_hip_rgcn_grouped_workspace_size(ctypes.addressof(rel), 1, n)
        ws = torch.empty(need, dtype=torch.uint8, device='cuda')
        rc = L.pyg_hip_rgcn_fused(3, x.data_ptr(), n, ctypes.addressof(rel), 1, out.data_ptr(), n, F, F, flags, ws.data_ptr(), need,
                                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return rc

    assert call(rows, cols, 1 | 8) == 0
    assert call(shuffled, cols, 1 | 8) != 0 and b'not grouped' in L.pyg_hip_last_error()
    assert call(bad_rows, cols, 1 | 8) != 0 and b'scatter index out of range' in L.pyg_hip_last_error()
    assert call(rows, bad_cols, 1 | 8) != 0 and b'gather index out of range' in L.pyg_hip_last_error()
    # too small a workspace is refused before anything is launched
    sd, gd = rows.cuda(), cols.cuda()
    rel = Rel(gd.data_ptr(), sd.data_ptr(), sd.numel(), 0, 0, w.data_ptr(), 0, 0, 0, 0)
    ws = torch.empty(256, dtype=torch.uint8, device='cuda')
    assert L.pyg_hip_rgcn_fused(3, x.data_ptr(), n, ctypes.addressof(rel), 1, out.data_ptr(), n, F, F, 8, ws.data_ptr(), 256,
                                torch.cuda.current_stream().cuda_stream) != 0
    assert b'workspace' in L.pyg_hip_last_error()


def test_grouped_feature_table_of_more_than_4_gib():
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(9)
    F = 128
    n_big = (1 << 24) + 4096
    free, _ = torch.cuda.mem_get_info()
    if free < 3 * n_big * F * 2:
        pytest.skip('not enough device memory for a 4 GiB table')
    picks = torch.cat([1 + 50_000 * torch.randperm(300, generator=g), (1 << 24) + 1 + torch.randperm(4000, generator=g)[:300],
                       torch.tensor([0, (1 << 24) - 1, 1 << 24, n_big - 1])])
    table = torch.empty(n_big, F, dtype=torch.bfloat16, device='cuda')
    vals = torch.randint(-1, 2, (picks.numel(), F), generator=g).float()
    table[picks.cuda()] = vals.bfloat16().cuda()
    ets = [('a', 'r0', 'a'), ('a', 'r1', 'a')]
    W = signed_permutations(2, F, g)
    rows, cols = {}, {}
    for et, c in zip(ets, [5000, 77]):
        rows[et] = torch.sort(torch.randint(0, 40, (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, picks.numel(), (c,), generator=g).cuda()
    want = exact_want(picks.numel(), F, ets, rows, cols, vals, W, [0, 0], [0, 0])
    assert want.abs().max() <= 256
    y = rgcn.rgcn_layer_fused_tables({'a': table}, {'a': picks.cuda()}, ['a'], rows, cols, ets, W.bfloat16().cuda(), grouped=True)
    assert torch.equal(y.double().cpu(), want)
    del table


def test_grouped_is_differentiable_and_runs_in_deterministic_mode():
    from pyg_lib_amd import rgcn, diagnostics
    g = torch.Generator().manual_seed(31)
    n, F = 400, 128
    ets = [('a', 'r0', 'a'), ('a', 'r1', 'a')]
    rows = {et: torch.sort(torch.randint(0, 90, (1500,), generator=g)).values.cuda() for et in ets}
    cols = {et: torch.randint(0, n, (1500,), generator=g).cuda() for et in ets}
    off = rgcn.type_offsets({'a': n}, ['a'])
    x0 = torch.randn(n, F, generator=g).bfloat16().cuda()
    w0 = (torch.randn(2, F, F, generator=g) / F ** 0.5).bfloat16().cuda()
    grads = []
    for grouped in (False, True):
        x = x0.clone().requires_grad_()
        w = w0.clone().requires_grad_()
        y = rgcn.rgcn_layer_fused(x, off, rows, cols, ets, w, grouped=grouped)
        y.float().square().sum().backward()
        grads.append((y.detach(), x.grad, w.grad))
    scale = grads[0][0].float().abs().max().item()
    assert (grads[0][0].float() - grads[1][0].float()).abs().max().item() <= 3e-2 * scale
    for a, b in zip(grads[0][1:], grads[1][1:]):   # backward: the same kernels on slightly different dOut
        assert (a.float() - b.float()).abs().max().item() <= 5e-2 * a.float().abs().max().item()
    # deterministic mode: grouped stays on the fused kernel (it has no atomics), ungrouped takes the chain
    marker = diagnostics.last_accumulate_info()
    torch.use_deterministic_algorithms(True)
    try:
        yd = rgcn.rgcn_layer_fused(x0, off, rows, cols, ets, w0, grouped=True)
        assert torch.equal(yd, grads[1][0])
        assert diagnostics.last_accumulate_info() == marker
        yo = torch.ops.pyg.rgcn_fused(x0, [cols[e] for e in ets], [rows[e] for e in ets], [0, 0], [0, 0], w0, torch.empty_like(x0), True)
        assert torch.equal(yo, yd)
        # ... and so does training: the backward takes the atomic-free chain for dX (its scatter index is the forward's
        # gather index: not grouped), the weight gradient has no atomics anyway -- the same bits on every run
        runs = []
        for _ in range(2):
            x = x0.clone().requires_grad_()
            w = w0.clone().requires_grad_()
            rgcn.rgcn_layer_fused(x, off, rows, cols, ets, w, grouped=True).float().square().sum().backward()
            runs.append((x.grad.clone(), w.grad.clone()))
        assert diagnostics.last_accumulate_info() == marker
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
        for a, b in zip(grads[1][1:], runs[0]):
            assert (a.float() - b.float()).abs().max().item() <= 5e-2 * a.float().abs().max().item()
    finally:
        torch.use_deterministic_algorithms(False)


def test_backward_runs_without_float_atomics_and_is_bit_reproducible():
    """dX scatters through the forward's GATHER index (sampled neighbours, draw order: not grouped).  One stable sort per
    sample groups the edges by source row (rgcn._transposed_sample), and the atomic-free kernel serves the backward with
    the roles swapped and W^T (rgcn.set_dx_mode('grouped'); the default under torch.use_deterministic_algorithms(True)):
    no accumulating launch in forward or backward, two runs give the same bits, exact on integer data, and the sort is
    shared by the layers of a model."""
    from pyg_lib_amd import sampler, rgcn, diagnostics, ops
    from tests.test_rgcn_gpu import MAG_TYPES, MAG_ETS, build_graph
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(12)
    sizes = {'paper': 40_000, 'author': 60_000, 'institution': 900, 'field_of_study': 4_000}
    rp, cl = build_graph(rng, sizes, MAG_ETS, 12)
    out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in rp.items()}, {e: dev(v) for e, v in cl.items()},
                                         {'paper': dev(rng.permutation(sizes['paper'])[:1024].astype(np.int64))},
                                         {e: [15, 10] for e in MAG_ETS})
    row_d, col_d, node_d = out[0], out[1], out[2]
    off = rgcn.type_offsets({t: node_d[t].numel() for t in MAG_TYPES}, MAG_TYPES)
    F = 128
    g = torch.Generator().manual_seed(3)
    # integer data: x, dOut small integers, W signed permutations -> dX = sum of +-dOut entries (exact), dW small sums
    x0 = torch.randint(-2, 3, (off['__total__'], F), generator=g).bfloat16().cuda()
    W0 = signed_permutations(len(MAG_ETS), F, g).bfloat16().cuda()
    c = torch.randint(-1, 2, (off['__total__'], F), generator=g).bfloat16().cuda()
    ops.scatter_sum(torch.ones(4, 2, device='cuda'), torch.tensor([[0, 1], [1, 0], [0, 0], [1, 1]], device='cuda'), 0, None, 2)
    marker = diagnostics.last_accumulate_info()
    before_mode = rgcn.set_dx_mode('grouped')
    runs = []
    for rep in range(2):
        xg, wg = x0.clone().requires_grad_(), W0.clone().requires_grad_()
        y = rgcn.rgcn_layer_fused(xg, off, row_d, col_d, MAG_ETS, wg)
        assert rgcn.last_layer_path() == 'grouped'
        (y * c).sum().backward()
        runs.append((xg.grad.clone(), wg.grad.clone()))
    assert diagnostics.last_accumulate_info() == marker        # nothing accumulated through atomics, forward or backward
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    want = torch.zeros(off['__total__'], F, dtype=torch.float64, device='cuda')
    for i, (s, r, d) in enumerate(MAG_ETS):
        want.index_add_(0, col_d[(s, r, d)] + off[d], c[row_d[(s, r, d)] + off[s]].double() @ W0[i].double().t())
    assert 4 < want.abs().max().item() <= 256
    assert torch.equal(runs[0][0].double(), want)
    # a second layer on the same sample reuses the sort
    n_cached = len(rgcn._transposed)
    xg, wg = x0.clone().requires_grad_(), W0.clone().requires_grad_()
    (rgcn.rgcn_layer_fused(xg, off, row_d, col_d, MAG_ETS, wg) * c).sum().backward()
    assert len(rgcn._transposed) == n_cached and torch.equal(xg.grad, runs[0][0])
    # float data: against round 5's path (the atomic kernel with swapped roles) and the chain
    g2 = torch.Generator(device='cuda').manual_seed(4)
    xf = torch.randn(off['__total__'], F, device='cuda', generator=g2).bfloat16()
    Wf = (torch.randn(len(MAG_ETS), F, F, device='cuda', generator=g2) / F ** 0.5).bfloat16()
    cf = torch.randn(off['__total__'], F, device='cuda', generator=g2)
    grads = {}
    for mode in ('grouped', 'atomic', 'chain'):
        xg, wg = xf.clone().requires_grad_(), Wf.clone().requires_grad_()
        rgcn.set_dx_mode('atomic' if mode == 'atomic' else 'grouped')
        if mode == 'chain':
            y = rgcn.rgcn_layer(xg, off, row_d, col_d, MAG_ETS, wg)
        else:
            y = rgcn.rgcn_layer_fused(xg, off, row_d, col_d, MAG_ETS, wg)
        (y.float() * cf).sum().backward()
        grads[mode] = xg.grad.float()
    rgcn.set_dx_mode(before_mode)
    # 'auto' (the default): the atomic kernel, unless torch.use_deterministic_algorithms(True)
    assert before_mode == 'auto'
    xg = xf.clone().requires_grad_()
    (rgcn.rgcn_layer_fused(xg, off, row_d, col_d, MAG_ETS, Wf).float() * cf).sum().backward()
    assert 'pyg_hip_rgcn_fused' in diagnostics.last_accumulate_info()
    marker2 = diagnostics.last_accumulate_info()
    torch.use_deterministic_algorithms(True)
    try:
        det = []
        for _ in range(2):
            xg = xf.clone().requires_grad_()
            (rgcn.rgcn_layer_fused(xg, off, row_d, col_d, MAG_ETS, Wf).float() * cf).sum().backward()
            det.append(xg.grad.clone())
        assert diagnostics.last_accumulate_info() == marker2 and torch.equal(det[0], det[1])
        assert torch.equal(det[0].float(), grads['grouped'])
    finally:
        torch.use_deterministic_algorithms(False)
    scale = grads['chain'].abs().max().item()
    assert (grads['grouped'] - grads['chain']).abs().max().item() <= 3e-2 * scale
    assert (grads['grouped'] - grads['atomic']).abs().max().item() <= 5e-2 * scale


@pytest.mark.parametrize('dtype,K,M', [(torch.bfloat16, 96, 40), (torch.float16, 64, 64), (torch.bfloat16, 256, 24), (torch.float32, 64, 32)])
def test_backward_of_other_widths_is_atomic_free_too(dtype, K, M):
    """The run-time-size instances serve the backward as well: dX[g_e] += dOut[s_e] @ W_r^T is the same kernel with the roles
    swapped (K and M exchanged) on the transposed sample (rgcn.set_dx_mode('grouped')).  No accumulating launch in forward or
    backward, two runs give the same bits, dX exact on integer data, dW against float64."""
    from pyg_lib_amd import rgcn, diagnostics, ops
    g = torch.Generator().manual_seed(31 + K + M)
    types = ['a', 'b']
    n = {'a': 700, 'b': 260}
    ets = [('a', 'r0', 'a'), ('b', 'r1', 'a'), ('a', 'r2', 'b')]
    rows, cols = {}, {}
    for et, c in zip(ets, [4000, 900, 2500]):
        s, _, d = et
        rows[et] = torch.sort(torch.randint(0, n[s] // 2, (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, n[d], (c,), generator=g).cuda()
    off = rgcn.type_offsets(n, types)
    x0 = torch.randint(-2, 3, (off['__total__'], K), generator=g).to(dtype).cuda()
    W0 = column_selectors(len(ets), K, M, g).to(dtype).cuda()
    c0 = torch.randint(-1, 2, (off['__total__'], M), generator=g).to(dtype).cuda()
    ops.scatter_sum(torch.ones(4, 2, device='cuda'), torch.tensor([[0, 1], [1, 0], [0, 0], [1, 1]], device='cuda'), 0, None, 2)
    marker = diagnostics.last_accumulate_info()
    before = rgcn.set_dx_mode('grouped')
    try:
        runs = []
        for _ in range(2):
            xg, wg = x0.clone().requires_grad_(), W0.clone().requires_grad_()
            y = rgcn.rgcn_layer_fused(xg, off, rows, cols, ets, wg, grouped=True)
            assert rgcn.last_layer_path() == 'grouped'
            (y * c0).sum().backward()
            runs.append((xg.grad.clone(), wg.grad.clone()))
        assert diagnostics.last_accumulate_info() == marker
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    finally:
        rgcn.set_dx_mode(before)
    want_x = torch.zeros(off['__total__'], K, dtype=torch.float64, device='cuda')
    want_w = torch.zeros(len(ets), K, M, dtype=torch.float64, device='cuda')
    for i, (s, r, d) in enumerate(ets):
        gi, si = cols[(s, r, d)] + off[d], rows[(s, r, d)] + off[s]
        want_x.index_add_(0, gi, c0[si].double() @ W0[i].double().t())
        want_w[i] = x0[gi].double().t() @ c0[si].double()
    assert want_x.abs().max().item() <= 256 and torch.equal(runs[0][0].double(), want_x)
    tol = {torch.float32: 1e-5, torch.bfloat16: 8e-3, torch.float16: 1.5e-3}[dtype] * max(want_w.abs().max().item(), 1.0)
    assert (runs[0][1].double() - want_w).abs().max().item() <= tol


def test_gather_index_beyond_32_bits_is_reported_not_aliased():
    """the 32-bit row arithmetic of tables below 4 GiB must not let an index of 2^32 + k pass as row k."""
    from pyg_lib_amd import rgcn
    et = ('a', 'r', 'a')
    n, F = 64, 128
    x = torch.ones(n, F, dtype=torch.bfloat16, device='cuda')
    w = torch.eye(F, dtype=torch.bfloat16, device='cuda')[None]
    rows = torch.tensor([0, 0, 1], device='cuda')
    cols = torch.tensor([3, (1 << 32) + 5, 7], device='cuda')
    off = rgcn.type_offsets({'a': n}, ['a'])
    torch.cuda.synchronize()
    rgcn.pending_index_error()
    y = rgcn.rgcn_layer_fused(x, off, {et: rows}, {et: cols}, [et], w, grouped=True)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 1
    assert y.shape == (n, F)
