"""Secondary legs of bench.py: the other BASELINE configs and hot-path ops, each as one small JSON object with its
algorithmic bytes (SURVEY.md 8(d) formulas), the measured time per call (HIP events on the launch stream, inputs
resident in HBM) and the fraction of the 8 TB/s HBM peak that makes.  Rank 0 / one GPU unless stated otherwise.

Every leg is bounded (a few launches of a fixed synthetic workload) so that the default `python bench.py` run still
finishes within a few minutes.
"""
import math
import time

import torch

HBM_PEAK_GBS = 8000.0


def _event_ms(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _rate(alg_bytes, ms):
    gbps = alg_bytes / (ms * 1e-3) / 1e9
    return dict(alg_bytes=int(alg_bytes), ms=round(ms, 4), GBps=round(gbps, 1), frac=round(gbps / HBM_PEAK_GBS, 4))


# ---------------------------------------------------------------------------------------------------
# C4: grouped_matmul, 512 variable-size groups, K = M = 256, bf16 (BASELINE.json configs[3])
# ---------------------------------------------------------------------------------------------------

def c4_group_rows(num_groups=512, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.exp(torch.rand(num_groups, generator=g) * (math.log(65536.0) - math.log(256.0)) +
                     math.log(256.0)).long().tolist()


def leg_c4(device, rank, world, iters=10, F=256, dtype=torch.bfloat16):
    """The fixed 512-group job, relation-sharded over `world` ranks by greedy LPT on the row counts
    (pyg_lib_amd/sharding.py).  `compute_only`: every rank multiplies its groups, outputs left in its slot of the
    pool; `incl_allgather`: plus the in-place RCCL all_gather_into_tensor that completes the pool on every rank.
    Both are max-over-ranks times of the same fixed job (strong scaling)."""
    import torch.distributed as dist
    from pyg_lib_amd import ops, sharding
    rows = c4_group_rows()
    plan = sharding.GroupPlan(rows, world)
    mine = plan.local_groups(rank)
    gd = torch.Generator(device=device).manual_seed(10 + rank)
    xs = [torch.randn(rows[i], F, device=device, generator=gd).to(dtype) for i in mine]
    ws = [(torch.randn(F, F, device=device, generator=gd) / F ** 0.5).to(dtype) for _ in mine]
    esz = xs[0].element_size()
    total_rows = sum(rows)
    flops = 2.0 * total_rows * F * F
    alg_total = esz * (2 * total_rows * F + len(rows) * F * F)   # whole job
    alg_local = esz * (2 * plan.load[rank] * F + len(mine) * F * F)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / n], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) * 1e3

    # the reference's operator signature (list in, list out): what a PyG caller runs on one GPU
    op_ms = timed(lambda: ops.grouped_matmul(xs, ws), iters)
    variant = ops.matmul_last_variant()
    kernel_ms = _event_ms(lambda: sharding.grouped_matmul_sharded(xs, ws, plan, rank, gather=False), iters)
    comp_ms = timed(lambda: sharding.grouped_matmul_sharded(xs, ws, plan, rank, gather=False), iters)
    res = dict(workload='grouped_matmul: 512 groups, rows log-uniform [256, 65536], K=M=256 (BASELINE.json configs[3])',
               groups=len(rows), rows=total_rows, F=F, dtype='bf16', n_gpus=world, kernel=variant,
               sharding=f'LPT by rows over {world} ranks, imbalance {plan.imbalance:.4f}',
               operator_ms=round(op_ms, 4),
               compute_only=dict(ms=round(comp_ms, 4), GFLOPs=round(flops / (comp_ms * 1e-3) / 1e9, 1),
                                 **{k: v for k, v in _rate(alg_total, comp_ms).items() if k != 'ms'}),
               rank0_launch=_rate(alg_local, kernel_ms))
    if world > 1:
        ag_ms = timed(lambda: sharding.grouped_matmul_sharded(xs, ws, plan, rank, gather=True), max(3, iters // 3))
        res['incl_allgather'] = dict(ms=round(ag_ms, 4), GFLOPs=round(flops / (ag_ms * 1e-3) / 1e9, 1),
                                     bytes_received_per_rank=int(esz * (total_rows - plan.load[rank]) * F))
    return res


# ---------------------------------------------------------------------------------------------------
# grouped_matmul with per-group shapes (HeteroDictLinear: one K per node type) and segment_matmul K = 100
# ---------------------------------------------------------------------------------------------------

def _kernel_ms(fn, iters=10, warmup=3):
    """Mean duration of the matmul kernel itself: HIP events recorded around it on its launch stream
    (pyg_hip_profile_*, include/pyg_hip.h)."""
    import ctypes
    from pyg_lib_amd import _capi
    L = _capi.lib()
    L.pyg_hip_profile_enable.argtypes = [ctypes.c_int]
    L.pyg_hip_profile_collect.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.pyg_hip_profile_collect.restype = ctypes.c_int
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    L.pyg_hip_profile_enable(1)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_float * iters)()
    n = L.pyg_hip_profile_collect(buf, iters)
    L.pyg_hip_profile_enable(0)
    return sum(buf[i] for i in range(min(n, iters))) / max(min(n, iters), 1)


def leg_grouped_mixed(device, rows_total=6_000_000, G=64, ks=(100, 128, 256, 768), M=128, dtype=torch.bfloat16):
    """64 groups, rows log-uniform [1 Ki, 256 Ki] scaled to 6 M rows, K cycling through 100 / 128 / 256 / 768, M = 128:
    a mixed-shape list as the reference's grouped GEMM takes (one GemmCoord per group,
    ops/cuda/matmul_kernel.cu:33-67).  Runs the general-shape MFMA kernel (csrc/hip/matmul_gen.hip); the
    one-thread-per-output kernel such lists ran before round 3 is timed beside it."""
    from pyg_lib_amd import ops
    g = torch.Generator().manual_seed(0)
    rows = torch.exp(torch.rand(G, generator=g) * (math.log(262144.0) - math.log(1024.0)) + math.log(1024.0))
    rows = (rows / rows.sum() * rows_total).long().tolist()
    gd = torch.Generator(device=device).manual_seed(1)
    kk = [ks[i % len(ks)] for i in range(G)]
    xs = [torch.randn(r, k, device=device, generator=gd).to(dtype) for r, k in zip(rows, kk)]
    ws = [(torch.randn(k, M, device=device, generator=gd) / k ** 0.5).to(dtype) for k in kk]
    s = xs[0].element_size()
    alg = sum(s * (r * k + r * M + k * M) for r, k in zip(rows, kk))
    flops = sum(2.0 * r * k * M for r, k in zip(rows, kk))
    ms = _kernel_ms(lambda: ops.grouped_matmul(xs, ws))
    variant = ops.matmul_last_variant()
    res = _rate(alg, ms)
    res.update(workload=f'grouped_matmul: {G} groups, {sum(rows)} rows, K in {list(ks)} -> M={M}, bf16 (per-group shapes)',
               kernel=variant, TFLOPs=round(flops / (ms * 1e-3) / 1e12, 1), bound='hbm')
    ops.set_matmul_schedule('naive')
    try:
        ms_naive = _kernel_ms(lambda: ops.grouped_matmul(xs, ws), iters=2, warmup=1)
        res['naive_kernel'] = dict(ms=round(ms_naive, 3), GBps=round(alg / (ms_naive * 1e-3) / 1e9, 1),
                                   slowdown=round(ms_naive / ms, 1))
    finally:
        ops.set_matmul_schedule('auto')
    del xs, ws
    # segment_matmul with ogbn-products' feature width
    N, K, B = 8_000_000, 100, 47
    x = torch.randn(N, K, device=device, generator=gd).to(dtype)
    w = (torch.randn(B, K, M, device=device, generator=gd) / 10).to(dtype)
    fr = torch.rand(B, generator=g)
    sizes = torch.floor(fr / fr.sum() * N).long()
    sizes[-1] += N - sizes.sum()
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    ms = _kernel_ms(lambda: ops.segment_matmul(x, ptr, w))
    res['segment_k100'] = dict(workload=f'segment_matmul: {N} rows, K=100 -> M={M}, {B} segments, bf16',
                               kernel=ops.matmul_last_variant(), **_rate(s * (N * K + N * M + B * K * M), ms))
    # its weight gradient alone (dW[b] = X_b^T dY_b through the general-shape dW kernel, matmul_dw_gen.hip): reads X and dY
    # once, writes B x K x M
    xg, wg = x.detach(), w.detach().requires_grad_()
    gy = torch.randn(N, M, device=device, generator=gd).to(dtype)
    before = ops.matmul_dw_counters()

    def dw_only():
        out = ops.segment_matmul(xg, ptr, wg)
        (gw,) = torch.autograd.grad(out, [wg], gy)
        return gw

    ms_fdw = _event_ms(dw_only, 5)
    after = ops.matmul_dw_counters()
    ms_f = _event_ms(lambda: ops.segment_matmul(x, ptr, w), 5)
    alg_dw = s * (N * K + N * M) + s * B * K * M
    r = _rate(alg_dw, max(ms_fdw - ms_f, 1e-6))
    r.update(workload=f'weight gradient of the K=100 segment_matmul above (one launch; forward time subtracted)',
             kernel='dw_gen_kernel' if after[1] > before[1] else ('seg_dw_kernel' if after[0] > before[0] else 'reference loop'),
             ms_forward_plus_dw=round(ms_fdw, 4))
    res['segment_k100_backward'] = r
    return res


def leg_segment_short(device, rows=1 << 22, seg_rows=256, F=128, dtype=torch.bfloat16):
    """Many short relations: 4 Mi rows cut into 256-row segments (16 384 relations, each with its own 32 KiB W -- a
    third of the bytes are weights).  The automatic choice is the item-ring kernel (a relation change = two ring items);
    the ticket kernel (W replicated in every wave's registers, refilled through a staging area) beside it."""
    from pyg_lib_amd import ops
    B = rows // seg_rows
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn(rows, F, device=device, generator=g).to(dtype)
    w = (torch.randn(B, F, F, device=device, generator=g) / F ** 0.5).to(dtype)
    ptr = torch.arange(0, rows + 1, seg_rows)
    esz = x.element_size()
    alg = 2 * rows * F * esz + B * F * F * esz + 8 * (B + 1)
    ms = _kernel_ms(lambda: ops.segment_matmul(x, ptr, w), iters=6, warmup=2)
    kern = ops.matmul_last_variant()
    try:
        ops.set_matmul_schedule('ticket')
        ms_t = _kernel_ms(lambda: ops.segment_matmul(x, ptr, w), iters=4, warmup=1)
        kern_t = ops.matmul_last_variant()
    finally:
        ops.set_matmul_schedule('auto')
    return dict(workload=f'segment_matmul, {rows} rows in {B} segments of {seg_rows}, F={F} bf16', kernel=kern,
                alg_bytes=int(alg), kernel_ms=round(ms, 4), GBps=round(alg / (ms * 1e-3) / 1e9, 1),
                frac=round(alg / (ms * 1e-3) / 8e12, 4), ticket=dict(kernel=kern_t, kernel_ms=round(ms_t, 4)))


def _clock_under(fn, ms_per_call, device, window_ms=40.0):
    """Average shader clock (MHz) over ~window_ms of back-to-back `fn` launches (pyg_hip_clock_probe on a side stream)."""
    import ctypes
    from pyg_lib_amd import _capi
    L = _capi.lib()
    L.pyg_hip_clock_probe.restype = ctypes.c_int
    L.pyg_hip_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    out = torch.zeros(2, dtype=torch.int64, device=device)
    side = torch.cuda.Stream(device)
    n = max(3, int(window_ms * 1.5 / max(ms_per_call, 1e-3)))
    for _ in range(2):
        fn()
    torch.cuda.synchronize(device)
    for _ in range(2):   # the probe starts once the load is running
        fn()
    _capi.check(L.pyg_hip_clock_probe(out.data_ptr(), float(window_ms), side.cuda_stream))
    for _ in range(n):
        fn()
    torch.cuda.synchronize(device)
    c, r = [int(v) for v in out.tolist()]
    return c / r * 100.0 if r > 0 else 0.0


def leg_segment_matmul_f32(device, make_c2, iters=5):
    """BASELINE configs[1] in fp32: north_star's 1e-5 parity configuration.  `exact` = torch's default precision
    ('highest'): IEEE fp32 MFMAs (AI = 32 flop/B is above the fp32 ridge of 157 TF / 8 TB/s = 20, so that kernel is bound by
    the fp32 matrix rate).  The top-level figures = torch.set_float32_matmul_precision('high'), the setting under which the
    reference switches to TF32: here split-bf16 (three bf16 terms per operand, six bf16 MFMAs per 16 k, fp32 accumulation --
    products exact to 2^-26, PYG_HIP_MM_F32_SPLIT in pyg_hip.h), whose matrix time is 2.7x lower: the bound is HBM."""
    from pyg_lib_amd import ops
    x, ptr, w, (N, B, F) = make_c2(device, 0, 1, torch.float32, 1.0)
    alg = 4 * (2 * N * F + B * F * F) + 8 * (B + 1)
    flop = 2.0 * N * F * F
    with ops.matmul_f32_split(False):  # torch's default precision ('highest')
        ms_e = _kernel_ms(lambda: ops.segment_matmul(x, ptr, w), iters=iters, warmup=2)
        kern_e = ops.matmul_last_variant()
        # the shader clock WHILE this kernel runs: a one-wave probe on a side stream watches the shader-clock counter against
        # the constant 100 MHz counter over a window of back-to-back launches (the data-sheet peak is quoted at 2.4 GHz)
        clock_mhz = _clock_under(lambda: ops.segment_matmul(x, ptr, w), ms_e, device)
    with ops.matmul_f32_split(True):   # torch.set_float32_matmul_precision('high')
        ms = _kernel_ms(lambda: ops.segment_matmul(x, ptr, w), iters=iters, warmup=2)
        kern_s = ops.matmul_last_variant()
    tf, tf_e = flop / (ms * 1e-3) / 1e12, flop / (ms_e * 1e-3) / 1e12
    gbps = alg / (ms * 1e-3) / 1e9
    return dict(workload='segment_matmul C2 in fp32 (154 relations, 21,111,007 rows, F=128)', kernel=kern_s,
                precision="torch float32_matmul_precision 'high' (split-bf16); `exact` = 'highest', torch's default",
                bound='hbm', achieved=round(gbps, 1), peak=8000.0, unit='GB/s', frac=round(gbps / 8000.0, 4),
                kernel_ms=round(ms, 4), alg_bytes=int(alg), tflops=round(tf, 1), vs_f32_mfma_peak=round(tf / 157.0, 4),
                exact=dict(kernel=kern_e, bound='mfma', achieved=round(tf_e, 1), peak=157.0, unit='TFLOP/s',
                           frac=round(tf_e / 157.0, 4), kernel_ms=round(ms_e, 4), clock_MHz_under_load=round(clock_mhz, 0),
                           peak_at_that_clock=round(157.0 * clock_mhz / 2400.0, 1),
                           frac_of_peak_at_that_clock=round(tf_e / (157.0 * clock_mhz / 2400.0), 4) if clock_mhz > 0 else None))


# ---------------------------------------------------------------------------------------------------
# C5: hetero_neighbor_sample + R-GCN layer on a MAG-shaped graph (BASELINE.json configs[4])
# ---------------------------------------------------------------------------------------------------

MAG_SIZES = {'paper': 736_389, 'author': 1_134_649, 'institution': 8_740, 'field_of_study': 59_965}
MAG_RELS = [('paper', 'cites', 'paper', 10_832_542), ('author', 'writes', 'paper', 7_145_660),
            ('paper', 'rev_writes', 'author', 7_145_660), ('author', 'affiliated_with', 'institution', 1_043_998),
            ('institution', 'rev_affiliated_with', 'author', 1_043_998),
            ('paper', 'has_topic', 'field_of_study', 7_505_078), ('field_of_study', 'rev_has_topic', 'paper', 7_505_078)]


def make_mag_graph(device, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    rp, cl = {}, {}
    for s, r, d, e in MAG_RELS:
        src = torch.randint(0, MAG_SIZES[s], (e,), device=device, generator=g)
        deg = torch.bincount(src, minlength=MAG_SIZES[s])
        rp[(s, r, d)] = torch.cat([deg.new_zeros(1), deg.cumsum(0)])
        cl[(s, r, d)] = torch.randint(0, MAG_SIZES[d], (e,), device=device, generator=g)
    return rp, cl


def make_mag_graph_csc(device, seed=0):
    """The same edge counts as CSC (pointer over the DST nodes, values = SRC ids): what the reference's own MAG benchmark
    feeds hetero_neighbor_sample(csc=True) (benchmark/sampler/hetero_neighbor.py:106-124)."""
    g = torch.Generator(device=device).manual_seed(seed + 7)
    cp, rw = {}, {}
    for s, r, d, e in MAG_RELS:
        dst = torch.randint(0, MAG_SIZES[d], (e,), device=device, generator=g)
        deg = torch.bincount(dst, minlength=MAG_SIZES[d])
        cp[(s, r, d)] = torch.cat([deg.new_zeros(1), deg.cumsum(0)])
        rw[(s, r, d)] = torch.randint(0, MAG_SIZES[s], (e,), device=device, generator=g)
    return cp, rw


def leg_c5_csc(device, feat, W, seeds, iters, F, esz):
    """C5 as the reference's benchmark runs it: csc=True.  `col` holds the expanded (dst-typed) nodes -- nondecreasing, so
    the layer takes the atomic-free kernel by default (grouped=None resolves through the sampler's registry)."""
    from pyg_lib_amd import sampler, rgcn
    types = list(MAG_SIZES)
    ets = [(s, r, d) for s, r, d, _ in MAG_RELS]
    cp, rw = make_mag_graph_csc(device)
    fan = {e: [15, 10] for e in ets}

    def one(i):
        out = sampler.hetero_neighbor_sample(cp, rw, {'paper': seeds[i]}, fan, csc=True)
        return out, rgcn.rgcn_layer_fused_tables(feat, out[2], types, out[0], out[1], ets, W, csc=True)

    for i in range(3):
        out, _ = one(i)
    path = rgcn.last_layer_path()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3, 3 + iters):
        out, _ = one(i)
    torch.cuda.synchronize()
    total_ms = (time.perf_counter() - t0) / iters * 1e3
    t0 = time.perf_counter()
    for i in range(3, 3 + iters):
        sampler.hetero_neighbor_sample(cp, rw, {'paper': seeds[i]}, fan, csc=True)
    torch.cuda.synchronize()
    samp_ms = (time.perf_counter() - t0) / iters * 1e3
    layer_ms = _event_ms(lambda: rgcn.rgcn_layer_fused_tables(feat, out[2], types, out[0], out[1], ets, W, csc=True), iters)
    torch.cuda.synchronize()
    e = sum(v.numel() for v in out[0].values())
    n = sum(v.numel() for v in out[2].values())
    return dict(what='the same step with csc=True (CSC graph: pointer over dst nodes; col = expanded nodes, row = sampled '
                     'neighbours; out[col] += x[row] @ W_r)', layer_path=path, layer_index_check=rgcn.pending_index_error(),
                sampler_mode=sampler.last_mode(), edges_last_batch=e, nodes_last_batch=n, ms_end_to_end=round(total_ms, 4),
                ms_sampler=round(samp_ms, 4), layer=_rate(e * (F * esz + 16) + n * F * esz + len(ets) * F * F * esz, layer_ms))


def leg_c5(device, iters=10, batch=1024, F=128, dtype=torch.bfloat16, grouped=True):
    from pyg_lib_amd import sampler, rgcn
    types = list(MAG_SIZES)
    ets = [(s, r, d) for s, r, d, _ in MAG_RELS]
    rp, cl = make_mag_graph(device)
    feat = {t: torch.randn(MAG_SIZES[t], F, device=device).to(dtype) for t in types}
    W = (torch.randn(len(ets), F, F, device=device) / F ** 0.5).to(dtype)
    fan = {e: [15, 10] for e in ets}
    gs = torch.Generator().manual_seed(1)
    seeds = [torch.randperm(MAG_SIZES['paper'], generator=gs)[:batch].to(device) for _ in range(iters + 3)]
    state = {}

    def sample(i):
        return sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds[i]}, fan)

    def layer(out):
        # features are gathered from the global tables inside the kernel: no per-batch feature matrix
        # (grouped: the sampler emits every relation's edges grouped by the node they were sampled for -- the atomic-free
        # owner-computes kernel, its promise verified on the device; `pending` below)
        return rgcn.rgcn_layer_fused_tables(feat, out[2], types, out[0], out[1], ets, W, grouped=grouped)

    def one(i):
        out = sample(i)
        y = layer(out)
        state['last'] = out
        return out, y

    torch.manual_seed(100)
    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = 0
    for i in range(3, 3 + iters):
        out, y = one(i)
        edges += sum(v.numel() for v in out[0].values())
    torch.cuda.synchronize()
    total_ms = (time.perf_counter() - t0) / iters * 1e3
    t0 = time.perf_counter()
    for i in range(3, 3 + iters):
        sample(i)
    torch.cuda.synchronize()
    samp_ms = (time.perf_counter() - t0) / iters * 1e3
    # K batches per call (hetero_neighbor_sample_batched: the batches' launch chains overlap on private streams; bit for bit
    # the single-batch results, tests/test_sampler_batched_gpu.py), each followed by its layer
    K = min(8, iters)
    sd = [{'paper': seeds[3 + k]} for k in range(K)]
    gseeds = [100 + k for k in range(K)]
    sampler.hetero_neighbor_sample_batched(rp, cl, sd, fan, gseeds)
    torch.cuda.synchronize()
    reps = max(4, iters // K)   # (a single call of K batches is a noisy sample)
    t0 = time.perf_counter()
    for _ in range(reps):
        sampler.hetero_neighbor_sample_batched(rp, cl, sd, fan, gseeds)
    torch.cuda.synchronize()
    samp_b_ms = (time.perf_counter() - t0) / (reps * K) * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        for o in sampler.hetero_neighbor_sample_batched(rp, cl, sd, fan, gseeds):
            layer(o)
    torch.cuda.synchronize()
    total_b_ms = (time.perf_counter() - t0) / (reps * K) * 1e3
    out = state['last']
    layer_ms = _event_ms(lambda: layer(out), iters)
    # C4's width on the same sample (informational; the atomic-free kernel takes K, M in {128, 256}, the atomic one 128 only):
    # against the three-op chain gather -> segment_matmul -> scatter_sum, which is what this width took before round 5
    F2 = 256
    feat2 = {t: torch.randn(MAG_SIZES[t], F2, device=device).to(dtype) for t in types}
    W2 = (torch.randn(len(ets), F2, F2, device=device) / F2 ** 0.5).to(dtype)
    f256_ms = _event_ms(lambda: rgcn.rgcn_layer_fused_tables(feat2, out[2], types, out[0], out[1], ets, W2, grouped=True), iters)
    f256_chain_ms = _event_ms(lambda: rgcn.rgcn_layer_fused_tables(feat2, out[2], types, out[0], out[1], ets, W2, grouped=False), max(2, iters // 3))
    del feat2
    # ... and float32 at F = 128 (plain FMAs in fp32 after the aggregation; the reference's 1e-5 configuration)
    feat4 = {t: torch.randn(MAG_SIZES[t], F, device=device) for t in types}
    W4 = torch.randn(len(ets), F, F, device=device) / F ** 0.5
    f32_ms = _event_ms(lambda: rgcn.rgcn_layer_fused_tables(feat4, out[2], types, out[0], out[1], ets, W4, grouped=True), iters)
    f32_chain_ms = _event_ms(lambda: rgcn.rgcn_layer_fused_tables(feat4, out[2], types, out[0], out[1], ets, W4, grouped=False), max(2, iters // 3))
    del feat4
    # the reductions' dim_size (ops/scatter.cpp:156-160): only the EXPANDED nodes can receive anything -- rows behind them are
    # neither computed nor zero-filled (reported beside `layer`, with its own algorithmic bytes)
    expanded = {t: int(sum(out[4][t][:-1])) for t in types}
    trim_ms = _event_ms(lambda: rgcn.rgcn_layer_fused_tables(feat, out[2], types, out[0], out[1], ets, W, grouped=grouped,
                                                           num_out_rows=expanded), iters)
    n_trim = sum(expanded.values())
    # training step of the layer on the same sample: forward + backward (dX through the atomic-free kernel on the transposed
    # sample -- one stable sort per sample, shared by a model's layers --, dW through the weight-gradient kernel); against
    # round 5's backward (dX through the atomic kernel with swapped roles)
    off_b = rgcn.type_offsets({t: out[2][t].numel() for t in types}, types)
    xb = torch.cat([feat[t][out[2][t]] for t in types])
    cb = torch.randn(off_b['__total__'], F, device=device).to(dtype)

    def train_step():
        xg, wg = xb.detach().requires_grad_(), W.detach().requires_grad_()
        (rgcn.rgcn_layer_fused(xg, off_b, out[0], out[1], ets, wg, grouped=grouped) * cb).sum().backward()

    fb_atomic_ms = _event_ms(train_step, iters)   # (the default: dX through the atomic kernel)
    rgcn.set_dx_mode('grouped')
    fb_ms = _event_ms(train_step, iters)
    rgcn.set_dx_mode('auto')
    csc_leg = leg_c5_csc(device, feat, W, seeds, iters, F, W.element_size())
    grouped_was = grouped
    grouped = not grouped_was
    other_ms = _event_ms(lambda: layer(out), iters)   # the other kernel (atomic adds into a zero-filled output / atomic-free)
    grouped = grouped_was
    torch.cuda.synchronize()
    pending = rgcn.pending_index_error()
    e = sum(v.numel() for v in out[0].values())
    n = sum(v.numel() for v in out[2].values())
    esz = W.element_size()
    # layer: per edge one gathered source row in (+ 16 B of indices), per node one output row written
    alg = e * (F * esz + 16) + n * F * esz + len(ets) * F * F * esz
    return dict(workload='hetero_neighbor_sample + R-GCN layer, MAG-shaped graph (4 node types, 7 relations, ~42 M '
                         'entries), batch 1024 papers, fanout [15, 10], F=128 bf16 (BASELINE.json configs[4])',
                layer_impl='rgcn_layer_fused_tables(grouped=%s)' % grouped, layer_index_check=pending,
                layer_other=dict(impl='grouped=%s' % (not grouped), ms=round(other_ms, 4)), sampler_mode=sampler.last_mode(), edges_per_batch=edges // iters, nodes_last_batch=n,
                ms_end_to_end=round(total_ms, 4), ms_sampler=round(samp_ms, 4), edges_per_s=round(edges / iters / (total_ms * 1e-3)),
                batched=dict(K=K, ms_sampler_per_batch=round(samp_b_ms, 4), ms_end_to_end_per_batch=round(total_b_ms, 4),
                             what='hetero_neighbor_sample_batched (K batches per call) + one fused layer per batch'),
                layer=_rate(alg, layer_ms),
                layer_trimmed=dict(_rate(e * (F * esz + 16) + n_trim * F * esz + len(ets) * F * F * esz, trim_ms), out_rows=n_trim,
                                   what='the same layer with num_out_rows = the expanded nodes per type (dim_size): the '
                                        'zero rows of the last hop\'s discoveries are not written'),
                backward=dict(ms_forward_backward=round(fb_atomic_ms, 4), ms_forward_backward_atomic_free=round(fb_ms, 4),
                              what='rgcn_layer_fused on the materialised batch features, loss = sum(y * c): forward + dX + '
                                   'dW + the autograd glue.  Default: dX through the atomic kernel with swapped roles; '
                                   'atomic_free: dX through the atomic-free kernel on the transposed sample (one stable '
                                   'sort per sample, cached) -- no float atomic anywhere in the step, bit-reproducible; the '
                                   'default under torch.use_deterministic_algorithms(True)'),
                csc=csc_leg,
                layer_f256=dict(_rate(e * (F2 * esz + 16) + n * F2 * esz + len(ets) * F2 * F2 * esz, f256_ms),
                                what='the same sample with F = 256 (rgcn_layer_fused_tables, grouped=True)',
                                three_op_chain_ms=round(f256_chain_ms, 4)),
                layer_f32=dict(_rate(e * (F * 4 + 16) + n * F * 4 + len(ets) * F * F * 4, f32_ms),
                               what='the same sample in float32, F = 128 (rgcn_layer_fused_tables, grouped=True)',
                               three_op_chain_ms=round(f32_chain_ms, 4)))


# ---------------------------------------------------------------------------------------------------
# index_sort, scatter_sum, segment_matmul backward
# ---------------------------------------------------------------------------------------------------

def leg_index_sort(device, n=100_000_000, max_value=2_449_029, iters=5):
    from pyg_lib_amd import ops
    g = torch.Generator(device=device).manual_seed(3)
    keys = torch.randint(0, max_value, (n,), device=device, generator=g, dtype=torch.long)
    ms = _event_ms(lambda: ops.index_sort(keys, max_value), iters)
    passes = (max(max_value, 1).bit_length() + 7) // 8
    floor = n * (8 + 16)                 # read keys once, write keys + permutation once
    res = _rate(floor, ms)
    res.update(workload=f'index_sort: {n} int64 keys < {max_value} ({passes} radix passes)', keys_per_s=round(n / (ms * 1e-3)),
               lsd_budget_bytes=int(passes * n * (8 + 16 + 16)))
    return res


def leg_scatter_sum(device, E=20_000_000, N=2_000_000, K=128, dtype=torch.bfloat16, iters=5):
    from pyg_lib_amd import ops
    g = torch.Generator(device=device).manual_seed(4)
    src = torch.randn(E, K, device=device, generator=g, dtype=torch.float32).to(dtype)
    idx = torch.randint(0, N, (E,), device=device, generator=g, dtype=torch.long)
    ms = _event_ms(lambda: ops.scatter_sum(src, idx, dim=0, dim_size=N), iters)
    esz = src.element_size()
    res = _rate(8 * E + esz * E * K + esz * N * K, ms)
    res.update(workload=f'scatter_sum: E={E} random (unsorted) indices into N={N} rows, K={K} bf16')
    idx_sorted = idx.sort().values
    ms2 = _event_ms(lambda: ops.segment_sum_coo(src, idx_sorted, dim_size=N), iters)
    res['segment_sum_coo_sorted'] = _rate(8 * E + esz * E * K + esz * N * K, ms2)
    ms3 = _event_ms(lambda: ops.gather_coo(src[:N], idx_sorted), iters)
    res['gather_coo'] = _rate(8 * E + 2 * esz * E * K, ms3)
    # the same call on a power-law-shaped index: ONE destination collects 2.5 % of the edges (a hub row: csr.hip deals its
    # chunks to all workgroups; round 6 start: 99 ms for the shape of tools/hub_sweep.py)
    hub = idx.clone()
    hub[torch.randperm(E, device=device, generator=g)[:E // 40]] = N // 3
    ms4 = _event_ms(lambda: ops.scatter_sum(src, hub, dim=0, dim_size=N), iters)
    res['hub_2_5_percent'] = dict(_rate(8 * E + esz * E * K + esz * N * K, ms4), slowdown_vs_uniform=round(ms4 / ms, 3))
    hub_sorted = hub.sort().values
    ms5 = _event_ms(lambda: ops.segment_sum_coo(src, hub_sorted, dim_size=N), iters)
    res['hub_2_5_percent']['segment_sum_coo_sorted_ms'] = round(ms5, 4)
    return res


def leg_backward(device, x, ptr, w, iters=5):
    """segment_matmul forward + backward (dX through the forward kernel with W^T read in place, dW through the
    split-rows weight-gradient kernel) on the bench's own C2 tensors."""
    from pyg_lib_amd import ops
    N, F = x.shape
    B = w.size(0)
    esz = x.element_size()
    xg = x.detach().requires_grad_()
    wg = w.detach().requires_grad_()
    go = torch.ones(N, F, device=device, dtype=x.dtype)

    def fb():
        out = ops.segment_matmul(xg, ptr, wg)
        out.backward(go)
        xg.grad = None
        wg.grad = None

    ms_fb = _event_ms(fb, iters)
    ms_f = _event_ms(lambda: ops.segment_matmul(x, ptr, w), iters)
    # backward alone: dX reads dY + writes dX, dW reads X + dY and writes fp32 accumulators + B*K*M
    alg_bwd = esz * (2 * N * F) + esz * (2 * N * F) + 4 * B * F * F
    res = _rate(alg_bwd, ms_fb - ms_f)
    res.update(workload='segment_matmul backward (dX + dW) on C2', ms_forward=round(ms_f, 4), ms_forward_backward=round(ms_fb, 4),
               flops=2 * 2.0 * N * F * F)
    return res
