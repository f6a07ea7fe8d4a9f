"""``csc=True`` through every form of the R-GCN layer on the device -- the mode of the reference's own MAG benchmark
(benchmark/sampler/hetero_neighbor.py:106-124: colptr_dict / row_dict, csc=True) and of PyG's loaders.  For edge type
(src, rel, dst): ``row`` = sampled neighbours (src-typed), ``col`` = expanded nodes (dst-typed, nondecreasing)
(pyg_lib/csrc/sampler/cpu/neighbor_kernel.cpp:715-719, :147-159), and

    out[col + off[dst]] += x[row + off[src]] @ W_r.

MAG-shaped graph with four node types of UNEQUAL sizes: the device sampler against the oracle bit for bit, then the chain,
the fused layer and the fused layer on feature tables (+ gradients) against a float64 restatement -- exact on integer
data; the sampler's `col` vectors take the atomic-free kernel by default (diagnostics.last_accumulate_info)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_rgcn_gpu import MAG_TYPES, MAG_ETS          # noqa: E402
from tests.test_rgcn_grouped_gpu import signed_permutations   # noqa: E402

SIZES = {'paper': 40_000, 'author': 60_000, 'institution': 900, 'field_of_study': 4_000}


def build_csc_graph(rng, sizes, ets, mean_deg):
    """One CSC per edge type (src, rel, dst): pointer over the DST nodes, values = SRC ids."""
    cp, rw = {}, {}
    for (s, r, d) in ets:
        deg = rng.poisson(mean_deg, sizes[d]).astype(np.int64)
        cp[(s, r, d)] = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        rw[(s, r, d)] = rng.integers(0, sizes[s], int(deg.sum()), dtype=np.int64)
    return cp, rw


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def sample_csc(seed=9, batch=1024, fan=(15, 10)):
    import oracle
    from pyg_lib_amd import sampler
    rng = np.random.default_rng(3)
    cp, rw = build_csc_graph(rng, SIZES, MAG_ETS, 12)
    seeds = {'paper': rng.permutation(SIZES['paper'])[:batch].astype(np.int64)}
    fans = {e: list(fan) for e in MAG_ETS}
    torch.manual_seed(seed)
    out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in cp.items()}, {e: dev(v) for e, v in rw.items()},
                                         {k: dev(v) for k, v in seeds.items()}, fans, csc=True)
    ref = oracle.hetero_neighbor_sample(MAG_TYPES, MAG_ETS, cp, rw, seeds, fans, csc=True, rng_seed=seed)
    return out, ref


def restate_csc(x, off, row_d, col_d, W, rounded=None):
    """float64 on the device: out[col + off[dst]] += x[row + off[src]] @ W_r; `rounded`: the atomic-free kernel's two
    roundings (feature sum per relation to T, result to T)."""
    want = torch.zeros(off['__total__'], W.size(2), dtype=torch.float64, device=x.device)
    for i, (s, r, d) in enumerate(MAG_ETS):
        agg = torch.zeros(off['__total__'], x.size(1), dtype=torch.float64, device=x.device)
        agg.index_add_(0, col_d[(s, r, d)] + off[d], x[row_d[(s, r, d)] + off[s]].double())
        if rounded is not None:
            agg = agg.to(rounded).double()
        want += agg @ W[i].double()
    return want


def test_csc_sample_is_bit_exact_and_its_roles_are_what_the_reference_defines():
    from pyg_lib_amd import sampler
    out, ref = sample_csc()
    row_d, col_d, node_d = out[0], out[1], out[2]
    for t in MAG_TYPES:
        assert torch.equal(node_d[t].cpu(), torch.from_numpy(ref[2][t])), t
    nn = {t: node_d[t].numel() for t in MAG_TYPES}
    assert len(set(nn.values())) == 4
    for e in MAG_ETS:
        assert torch.equal(row_d[e].cpu(), torch.from_numpy(ref[0][e])) and torch.equal(col_d[e].cpu(), torch.from_numpy(ref[1][e])), e
        assert torch.equal(out[3][e].cpu(), torch.from_numpy(ref[3][e])), e
        assert bool((col_d[e][1:] >= col_d[e][:-1]).all()), e            # expanded nodes: grouped
        assert sampler.rows_are_grouped(col_d[e]) and not sampler.rows_are_grouped(row_d[e])
        if row_d[e].numel():   # (institutions are only discovered in the last hop: nothing is sampled FOR them)
            assert int(row_d[e].max()) < nn[e[0]] and int(col_d[e].max()) < nn[e[2]]
    assert sum(v.numel() > 0 for v in row_d.values()) >= 6
    assert sum(v.numel() for v in row_d.values()) > 100_000


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_csc_layers_are_exact_on_integer_data(dtype):
    """Small integers times signed-permutation weights: every message and every sum of <= 15 x 4 of them is exactly
    representable, so the chain, the atomic kernel, the atomic-free kernel and the tables variant must all equal the
    float64 restatement bit for bit."""
    from pyg_lib_amd import rgcn, diagnostics, ops
    out, _ = sample_csc()
    row_d, col_d, node_d = out[0], out[1], out[2]
    F = 128
    g = torch.Generator().manual_seed(6)
    feat = {t: torch.randint(-2, 3, (SIZES[t], F), generator=g).to(dtype).cuda() for t in MAG_TYPES}
    W = signed_permutations(len(MAG_ETS), F, g).to(dtype).cuda()
    off = rgcn.type_offsets({t: node_d[t].numel() for t in MAG_TYPES}, MAG_TYPES)
    x = torch.cat([feat[t][node_d[t]] for t in MAG_TYPES])
    want = restate_csc(x, off, row_d, col_d, W)
    assert 8 < want.abs().max().item() <= 256
    touched = torch.zeros(off['__total__'], dtype=torch.bool, device='cuda')
    for (s, r, d) in MAG_ETS:
        touched[col_d[(s, r, d)] + off[d]] = True
    assert 0 < int(touched.sum()) < off['__total__']

    y3 = rgcn.rgcn_layer(x, off, row_d, col_d, MAG_ETS, W, csc=True)
    assert torch.equal(y3.double(), want)
    # an atomic scatter in between, so that the marker is not a fused layer's
    ops.scatter_sum(torch.ones(4, 2, device='cuda'), torch.tensor([[0, 1], [1, 0], [0, 0], [1, 1]], device='cuda'), 0, None, 2)
    marker = diagnostics.last_accumulate_info()
    assert 'pyg_hip_rgcn_fused' not in marker
    y = rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, csc=True)                 # default: sampler `col` -> atomic-free
    yt = rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, W, csc=True)
    assert diagnostics.last_accumulate_info() == marker                                    # no accumulating launch ran
    assert rgcn.last_layer_path() == 'grouped'
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    assert torch.equal(y.double(), want) and torch.equal(yt.double(), want)
    assert not y[~touched].any()
    if dtype != torch.float32:   # (the atomic kernel is 16-bit only; float32 ungrouped is the chain)
        ya = rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, csc=True, grouped=False)
        assert 'pyg_hip_rgcn_fused' in diagnostics.last_accumulate_info()
        assert torch.equal(ya.double(), want)
        yta = rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, W, csc=True, grouped=False)
        assert torch.equal(yta.double(), want)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_csc_layers_on_float_data_and_gradients(dtype):
    from pyg_lib_amd import rgcn
    out, _ = sample_csc(seed=10, batch=512)
    row_d, col_d, node_d = out[0], out[1], out[2]
    F = 128
    g = torch.Generator(device='cuda').manual_seed(4)
    feat = {t: torch.randn(SIZES[t], F, device='cuda', generator=g).to(dtype) for t in MAG_TYPES}
    W = (torch.randn(len(MAG_ETS), F, F, device='cuda', generator=g) / F ** 0.5).to(dtype)
    off = rgcn.type_offsets({t: node_d[t].numel() for t in MAG_TYPES}, MAG_TYPES)
    x = torch.cat([feat[t][node_d[t]] for t in MAG_TYPES])
    want = restate_csc(x, off, row_d, col_d, W, rounded=dtype)
    scale = want.abs().max().item()
    y = rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, csc=True)
    assert (y.double() - want).abs().max().item() <= (8e-3 if dtype == torch.bfloat16 else 1.5e-3) * scale
    tol = (3e-2 if dtype == torch.bfloat16 else 4e-3) * scale
    for other in (rgcn.rgcn_layer(x, off, row_d, col_d, MAG_ETS, W, csc=True),
                  rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, csc=True, grouped=False),
                  rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, W, csc=True)):
        assert (y.double() - other.double()).abs().max().item() <= tol
    # gradients of sum(y * c) for a fixed random c: float64 autograd of the restatement vs the three layer forms
    c = torch.randn(off['__total__'], F, device='cuda', generator=g)
    x64 = x.double().requires_grad_()
    w64 = W.double().requires_grad_()
    acc = torch.zeros(off['__total__'], F, dtype=torch.float64, device='cuda')
    for i, (s, r, d) in enumerate(MAG_ETS):
        acc = acc.index_add(0, col_d[(s, r, d)] + off[d], x64[row_d[(s, r, d)] + off[s]] @ w64[i])
    (acc * c.double()).sum().backward()
    gx_s, gw_s = x64.grad.abs().max().item(), w64.grad.abs().max().item()
    for form in ('chain', 'fused', 'fused_atomic', 'tables'):
        xg = x.clone().requires_grad_()
        wg = W.clone().requires_grad_()
        if form == 'chain':
            yy = rgcn.rgcn_layer(xg, off, row_d, col_d, MAG_ETS, wg, csc=True)
        elif form == 'tables':
            fg = {t: feat[t].clone().requires_grad_() for t in MAG_TYPES}
            yy = rgcn.rgcn_layer_fused_tables(fg, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, wg, csc=True)
        else:
            yy = rgcn.rgcn_layer_fused(xg, off, row_d, col_d, MAG_ETS, wg, csc=True, grouped=None if form == 'fused' else False)
        (yy.float() * c).sum().backward()
        assert (wg.grad.double() - w64.grad).abs().max().item() <= 3e-2 * gw_s, form
        if form == 'tables':
            want_t = {t: torch.zeros(SIZES[t], F, dtype=torch.float64, device='cuda') for t in MAG_TYPES}
            for t in MAG_TYPES:
                want_t[t].index_add_(0, node_d[t], x64.grad[off[t]:off[t] + node_d[t].numel()])
                assert (fg[t].grad.double() - want_t[t]).abs().max().item() <= 3e-2 * gx_s, (form, t)
        else:
            assert (xg.grad.double() - x64.grad).abs().max().item() <= 3e-2 * gx_s, form


def test_a_sampler_output_modified_in_place_takes_the_atomic_kernel_and_gives_the_right_answer():
    """VERDICT r5 weak 2 / ADVICE r5 (medium): `grouped=None` must not trust a remembered tensor that was written to
    afterwards -- no deferred error 3, no silently wrong rows: the atomic kernel on the tensor as it is now."""
    from pyg_lib_amd import sampler, rgcn, diagnostics
    out, _ = sample_csc(seed=11, batch=256)
    row_d, col_d, node_d = out[0], out[1], out[2]
    F = 128
    g = torch.Generator().manual_seed(8)
    feat = {t: torch.randint(-2, 3, (SIZES[t], F), generator=g).bfloat16().cuda() for t in MAG_TYPES}
    W = signed_permutations(len(MAG_ETS), F, g).bfloat16().cuda()
    off = rgcn.type_offsets({t: node_d[t].numel() for t in MAG_TYPES}, MAG_TYPES)
    x = torch.cat([feat[t][node_d[t]] for t in MAG_TYPES])
    e0 = MAG_ETS[0]
    assert sampler.rows_are_grouped(col_d[e0])
    # reverse one relation's edge list in place (both vectors: the same edges, now in decreasing col order)
    col_d[e0].copy_(col_d[e0].flip(0))
    row_d[e0].copy_(row_d[e0].flip(0))
    assert not sampler.rows_are_grouped(col_d[e0]) and sampler.rows_are_grouped(col_d[MAG_ETS[1]])
    want = restate_csc(x, off, row_d, col_d, W)
    y = rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, csc=True)
    assert 'pyg_hip_rgcn_fused' in diagnostics.last_accumulate_info() and rgcn.last_layer_path() == 'atomic'
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    assert torch.equal(y.double(), want)
    # sort_ back into order: still "modified", still the atomic kernel, still right
    col_d[e0].copy_(col_d[e0].flip(0))
    row_d[e0].copy_(row_d[e0].flip(0))
    assert not sampler.rows_are_grouped(col_d[e0])
    assert torch.equal(rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, csc=True).double(), want)


def test_full_size_c5_csc_sample_is_bit_exact_and_layer_matches():
    """BASELINE config C5 at FULL size the way the reference's benchmark runs it (csc=True; the CSC graph of bench.py's
    `c5.csc` sub-leg: 1.94 M nodes of 4 types, 7 relations, 42.2 M entries; batch 1024 papers, fan-out [15, 10]): every
    sampler output against the oracle, then the variant the bench times (feature tables, default `grouped`) against a
    float64 restatement, bit-reproducible."""
    import oracle
    import bench_legs
    from pyg_lib_amd import sampler, rgcn
    types = list(bench_legs.MAG_SIZES)
    ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
    cp, rw = bench_legs.make_mag_graph_csc(torch.device('cuda:0'))
    fan = {e: [15, 10] for e in ets}
    seeds = torch.randperm(bench_legs.MAG_SIZES['paper'], generator=torch.Generator().manual_seed(1))[:1024]
    torch.manual_seed(2025)
    out = sampler.hetero_neighbor_sample(cp, rw, {'paper': seeds.cuda()}, fan, csc=True)
    assert sampler.last_mode() == 'fused'
    ref = oracle.hetero_neighbor_sample(types, ets, {e: v.cpu().numpy() for e, v in cp.items()},
                                        {e: v.cpu().numpy() for e, v in rw.items()}, {'paper': seeds.numpy()}, fan,
                                        csc=True, rng_seed=2025)
    row_d, col_d, node_d, edge_d = out[0], out[1], out[2], out[3]
    for e in ets:
        assert torch.equal(row_d[e].cpu(), torch.from_numpy(ref[0][e])), e
        assert torch.equal(col_d[e].cpu(), torch.from_numpy(ref[1][e])), e
        assert torch.equal(edge_d[e].cpu(), torch.from_numpy(ref[3][e])), e
        assert list(out[5][e]) == list(ref[5][e]), e
    for t in types:
        assert torch.equal(node_d[t].cpu(), torch.from_numpy(ref[2][t])), t
        assert list(out[4][t]) == list(ref[4][t]), t
    assert sum(v.numel() for v in row_d.values()) > 400_000
    F = 128
    g = torch.Generator(device='cuda').manual_seed(3)
    feat = {t: torch.randn(bench_legs.MAG_SIZES[t], F, device='cuda', generator=g).bfloat16() for t in types}
    W = (torch.randn(len(ets), F, F, device='cuda', generator=g) / F ** 0.5).bfloat16()
    off = rgcn.type_offsets({t: node_d[t].numel() for t in types}, types)
    x = torch.cat([feat[t][node_d[t]] for t in types])
    want = torch.zeros(off['__total__'], F, dtype=torch.float64, device='cuda')
    for i, (s, r, d) in enumerate(ets):
        agg = torch.zeros(off['__total__'], F, dtype=torch.float64, device='cuda')
        agg.index_add_(0, col_d[(s, r, d)] + off[d], x[row_d[(s, r, d)] + off[s]].double())
        want += agg.bfloat16().double() @ W[i].double()
    scale = want.abs().max().item()
    yt = rgcn.rgcn_layer_fused_tables(feat, node_d, types, row_d, col_d, ets, W, csc=True)
    assert rgcn.last_layer_path() == 'grouped'
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    assert scale > 1.0 and (yt.double() - want).abs().max().item() <= 8e-3 * scale
    assert torch.equal(yt, rgcn.rgcn_layer_fused_tables(feat, node_d, types, row_d, col_d, ets, W, csc=True))
    ya = rgcn.rgcn_layer_fused_tables(feat, node_d, types, row_d, col_d, ets, W, csc=True, grouped=False)
    assert (ya.double() - yt.double()).abs().max().item() <= 3e-2 * scale


@pytest.mark.parametrize('csc', [True, False])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_num_out_rows_trimmed_layer_equals_the_untrimmed_rows_bit_for_bit(csc, dtype):
    """`num_out_rows` (the reductions' dim_size, pyg_lib/csrc/ops/scatter.cpp:156-160) = the EXPANDED nodes per type: the
    trimmed result is the untrimmed one without its zero tail -- same bits (float data: the atomic-free kernel sums in the
    same order either way) -- through every form of the layer; the rows left out are zeros; gradients agree."""
    from pyg_lib_amd import sampler, rgcn
    from tests.test_rgcn_gpu import build_graph
    rng = np.random.default_rng(3)
    if csc:
        ptr, idx = build_csc_graph(rng, SIZES, MAG_ETS, 12)
    else:
        ptr, idx = build_graph(rng, SIZES, MAG_ETS, 12)
    seeds = {'paper': dev(rng.permutation(SIZES['paper'])[:1024].astype(np.int64))}
    out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in ptr.items()}, {e: dev(v) for e, v in idx.items()}, seeds,
                                         {e: [15, 10] for e in MAG_ETS}, csc=csc)
    row_d, col_d, node_d, nph = out[0], out[1], out[2], out[4]
    nn = {t: node_d[t].numel() for t in MAG_TYPES}
    expanded = {t: int(sum(nph[t][:-1])) for t in MAG_TYPES}
    assert 0 < sum(expanded.values()) < sum(nn.values()) // 3
    F = 128
    g = torch.Generator(device='cuda').manual_seed(4)
    feat = {t: torch.randn(SIZES[t], F, device='cuda', generator=g).to(dtype) for t in MAG_TYPES}
    W = (torch.randn(len(MAG_ETS), F, F, device='cuda', generator=g) / F ** 0.5).to(dtype)
    off = rgcn.type_offsets(nn, MAG_TYPES)
    ooff = rgcn.out_offsets(off, expanded)
    x = torch.cat([feat[t][node_d[t]] for t in MAG_TYPES])

    def rows_match(trim, full):
        assert trim.shape == (ooff['__total__'], F)
        for t in MAG_TYPES:
            assert torch.equal(trim[ooff[t]:ooff[t] + expanded[t]], full[off[t]:off[t] + expanded[t]]), t
            assert not full[off[t] + expanded[t]:off[t] + nn[t]].any(), t

    full = rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, csc=csc)
    assert rgcn.last_layer_path() == 'grouped'
    trim = rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, csc=csc, num_out_rows=expanded)
    assert rgcn.last_layer_path() == 'grouped'
    rows_match(trim, full)
    rows_match(rgcn.rgcn_layer_fused_tables(feat, node_d, MAG_TYPES, row_d, col_d, MAG_ETS, W, csc=csc, num_out_rows=expanded), full)
    rows_match(rgcn.rgcn_layer(x, off, row_d, col_d, MAG_ETS, W, csc=csc, num_out_rows=expanded),
               rgcn.rgcn_layer(x, off, row_d, col_d, MAG_ETS, W, csc=csc))
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 0
    # gradients: a loss on the trimmed rows = the same loss on the untrimmed result's first rows
    c = torch.randn(ooff['__total__'], F, device='cuda', generator=g)
    cf = torch.zeros(off['__total__'], F, device='cuda')
    for t in MAG_TYPES:
        cf[off[t]:off[t] + expanded[t]] = c[ooff[t]:ooff[t] + expanded[t]]
    grads = []
    for nrows, cc in ((expanded, c), (None, cf)):
        xg, wg = x.clone().requires_grad_(), W.clone().requires_grad_()
        (rgcn.rgcn_layer_fused(xg, off, row_d, col_d, MAG_ETS, wg, csc=csc, num_out_rows=nrows).float() * cc).sum().backward()
        grads.append((xg.grad, wg.grad))
    for a, b in zip(*grads):
        assert (a.float() - b.float()).abs().max().item() <= 5e-2 * b.float().abs().max().item()   # (bf16 dX: packed atomics, order-dependent rounding)
    # a count that is too small: reported like any scatter index out of range (error 2), nothing written out of bounds
    small = dict(expanded)
    small['paper'] = 5
    rgcn.rgcn_layer_fused(x, off, row_d, col_d, MAG_ETS, W, csc=csc, num_out_rows=small)
    torch.cuda.synchronize()
    assert rgcn.pending_index_error() == 2
