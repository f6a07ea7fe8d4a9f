"""Repetition under memory noise of the kernels whose waits are placed by hand.

The ticket kernel, the pipelined fp32 kernel, the 64-rows-per-wave K = 256 kernel and the fused R-GCN kernel issue
their loads from inline asm and wait with exact `vmcnt` counts; a miscounted wait reads a register before its load has
landed -- a bug that shows as ONE garbage tile in one of many runs, and only when the load happens to be late (commit
f4884f2 fixed such a count in the fp32 kernel for M = 64 / 32: the parity suite passed 14 times out of 15 with it).
Here every kernel runs 32 times on fresh ragged partitions while a second stream keeps the memory system busy with
large copies (late loads, uneven arrival), and every run must reproduce, bit for bit, a result that was itself checked:
the 16-bit K = M = 128 kernels against the contiguous-range kernel, the others against their own first run after that
run was compared with a float64 product.

Checked in round 3 by reverting f4884f2's wait counts in a scratch build: `test_fp32_pipelined_kernel_...[32]` failed
in 3 of 3 runs of this file, `[64]` in 2 of 3 (the unreverted build passes).
"""
import numpy as np
import pytest
import torch

from pyg_lib_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
REPS = 32


class Noise:
    """Large device copies on a side stream: ~100 us each, queued ahead so that they overlap whatever the main
    stream launches next."""

    def __init__(self, mbytes=192):
        self.stream = torch.cuda.Stream()
        self.a = torch.empty(mbytes << 20, dtype=torch.uint8, device=DEV).random_()
        self.b = torch.empty_like(self.a)

    def burst(self, n=24):
        with torch.cuda.stream(self.stream):
            for i in range(n):
                if i & 1:
                    self.a.copy_(self.b)
                else:
                    self.b.copy_(self.a)


def ragged_ptr(rng, B, kind, scale):
    if kind == 0:
        sizes = rng.integers(0, 3 * scale, B)
    elif kind == 1:
        sizes = rng.integers(0, 70, B)
    elif kind == 2:
        sizes = (rng.random(B) < 0.5) * rng.integers(1, 8 * scale, B)
    else:
        sizes = np.array([int(rng.integers(1, 120 * scale))] + [int(v) for v in rng.integers(0, 65, B - 1)])
    return torch.tensor([0] + np.cumsum(sizes).tolist())


@pytest.mark.parametrize('mode', ['ticket', 'ring'])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_ticket_kernel_under_noise_matches_contiguous_bitwise(dtype, mode):
    rng = np.random.default_rng(1)
    noise = Noise()
    g = torch.Generator(device=DEV).manual_seed(1)
    name = 'bf16' if dtype == torch.bfloat16 else 'f16'
    for rep in range(REPS):
        B = int(rng.integers(1, 200))
        ptr = ragged_ptr(rng, B, rep % 4, 1000)
        n = int(ptr[-1])
        if n == 0:
            continue
        x = torch.randn(n, 128, device=DEV, generator=g).to(dtype)
        w = (torch.randn(B, 128, 128, device=DEV, generator=g) / 11).to(dtype)
        bias = torch.randn(B, 128, device=DEV, generator=g).to(dtype) if rep % 3 == 0 else None
        try:
            ops.set_matmul_schedule('contiguous')
            ref = ops.segment_matmul(x, ptr, w, bias)
            ops.set_matmul_schedule(mode)
            noise.burst()
            out = ops.segment_matmul(x, ptr, w, bias)
            assert ops.matmul_last_variant() == f'mfma_{name}_k128_mc128_{mode}'
        finally:
            ops.set_matmul_schedule('auto')
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), (rep, B, n)


def _repeat_against_first(make_call, check_first, reps=REPS):
    noise = Noise()
    first = None
    for rep in range(reps):
        noise.burst()
        out = make_call()
        torch.cuda.synchronize()
        if first is None:
            check_first(out)
            first = [o.clone() for o in out] if isinstance(out, (list, tuple)) else out.clone()
        elif isinstance(out, (list, tuple)):
            for a, b in zip(out, first):
                assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int16),
                                   b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int16)), rep
        else:
            assert torch.equal(out.view(torch.int32) if out.dtype == torch.float32 else out.view(torch.int16),
                               first.view(torch.int32) if first.dtype == torch.float32 else first.view(torch.int16)), rep


@pytest.mark.parametrize('M', [128, 64, 32])
def test_fp32_pipelined_kernel_under_noise_is_reproducible(M):
    """M = 64 / 32 are the widths whose hand-kept store count was wrong before f4884f2."""
    rng = np.random.default_rng(M)
    g = torch.Generator(device=DEV).manual_seed(M)
    B = 61
    ptr = ragged_ptr(rng, B, 0, 1500)
    n = int(ptr[-1])
    x = torch.randn(n, 128, device=DEV, generator=g)
    w = torch.randn(B, 128, M, device=DEV, generator=g) / 11

    def check(out):
        assert ops.matmul_last_variant() == f'mfma_f32_k128_mc{M}'
        for b in (0, 7, 30, 60):
            s, e = int(ptr[b]), int(ptr[b + 1])
            ref = x[s:e].double() @ w[b].double()
            assert (out[s:e].double() - ref).norm() <= 1e-5 * max(ref.norm().item(), 1e-30)
        assert torch.isfinite(out).all()

    with ops.matmul_f32_split(False):  # (torch's default: the exact fp32 MFMA kernel)
        _repeat_against_first(lambda: ops.segment_matmul(x, ptr, w), check)


@pytest.mark.parametrize('M', [128, 256])
def test_fp32_split_bf16_kernel_under_noise_is_reproducible(M):
    """The split-bf16 kernel loads X through inline asm into AGPRs and waits with `vmcnt(16)` when exactly the 16
    unpredicated stores of a whole tile are younger; ragged segments mix whole and partial tiles."""
    rng = np.random.default_rng(M + 1)
    g = torch.Generator(device=DEV).manual_seed(M + 1)
    B = 61
    ptr = ragged_ptr(rng, B, 0, 1500)
    n = int(ptr[-1])
    x = torch.randn(n, 128, device=DEV, generator=g)
    w = torch.randn(B, 128, M, device=DEV, generator=g) / 11
    bias = torch.randn(B, M, device=DEV, generator=g)

    def check(out):
        assert ops.matmul_last_variant() == 'mfma_f32_k128_mc128_x3'
        for b in (0, 7, 30, 60):
            s, e = int(ptr[b]), int(ptr[b + 1])
            ref = x[s:e].double() @ w[b].double() + bias[b].double()
            assert (out[s:e].double() - ref).norm() <= 1e-6 * max(ref.norm().item(), 1e-30)
        assert torch.isfinite(out).all()

    with ops.matmul_f32_split(True):
        _repeat_against_first(lambda: ops.segment_matmul(x, ptr, w, bias), check)


def test_fp32_split_bf16_register_w_kernel_under_noise_is_reproducible():
    """The ring kernel's fp32 variant (`'ring'` forces it for any segment length): DMA from inline asm, waits naming
    4 - 12 younger operations, results staged in the ring slot that is refilled right behind the stores' LDS reads."""
    rng = np.random.default_rng(5)
    g = torch.Generator(device=DEV).manual_seed(5)
    B = 200
    ptr = ragged_ptr(rng, B, 1, 1)
    ptr = torch.cat([ptr, ptr[-1:] + torch.tensor([5000, 5001, 12000])])
    n = int(ptr[-1])
    x = torch.randn(n, 128, device=DEV, generator=g)
    w = torch.randn(B + 3, 128, 128, device=DEV, generator=g) / 11
    bias = torch.randn(B + 3, 128, device=DEV, generator=g)

    def check(out):
        assert ops.matmul_last_variant() == 'mfma_f32_k128_regw_x3'
        for b in (0, 7, 30, 199, 200, 202):
            s, e = int(ptr[b]), int(ptr[b + 1])
            ref = x[s:e].double() @ w[b].double() + bias[b].double()
            assert (out[s:e].double() - ref).norm() <= 1e-6 * max(ref.norm().item(), 1e-30)
        assert torch.isfinite(out).all()

    try:
        ops.set_matmul_schedule('ring')
        with ops.matmul_f32_split(True):
            _repeat_against_first(lambda: ops.segment_matmul(x, ptr, w, bias), check)
    finally:
        ops.set_matmul_schedule('auto')


@pytest.mark.parametrize('mode,variant', [('auto', 'mfma_bf16_k256_regw'), ('cyclic', 'mfma_bf16_k256_wide256r2')])
def test_k256_kernels_under_noise_are_reproducible(mode, variant):
    """K = M = 256: the register-W kernel (X tiles and W chunks by LDS-DMA from inline asm, waits that name 8 - 20 younger
    operations) and the 64-rows-per-wave kernel.  Many short relations: every workgroup changes relation often, with
    and without bias, plain and transposed weights."""
    rng = np.random.default_rng(256)
    g = torch.Generator(device=DEV).manual_seed(256)
    rows = [int(v) for v in rng.integers(0, 9000, 96)]
    rows[3] = 0
    rows[10] = 1
    rows[11] = 33
    rows[12] = 32
    rows[13] = 64
    xs = [torch.randn(r, 256, device=DEV, generator=g).bfloat16() for r in rows]
    ws = [(torch.randn(256, 256, device=DEV, generator=g) / 16).bfloat16() for _ in rows]
    ws = [w.t().contiguous().t() if i % 3 == 1 else w for i, w in enumerate(ws)]  # every third one stored [M][K]

    def check(outs):
        assert ops.matmul_last_variant() == variant
        for i in (0, 1, 5, 10, 11, 12, 13, 50, 95):
            ref = (xs[i].double() @ ws[i].double())
            torch.testing.assert_close(outs[i].double(), ref, rtol=2 ** -7, atol=2e-2)

    try:
        ops.set_matmul_schedule(mode)
        _repeat_against_first(lambda: ops.grouped_matmul(xs, ws), check)
    finally:
        ops.set_matmul_schedule('auto')


def test_k256_register_w_kernel_matches_the_lds_w_kernel_bitwise():
    """Random ragged partitions with bias (segment_matmul): the two kernels accumulate every output element in the
    same k order, so their results must be the same bits -- empty relations, single rows, 32 / 33 / 64 / 65-row
    relations (the empty and the partial second half of a 64-row tile), runs of tiny relations."""
    rng = np.random.default_rng(77)
    g = torch.Generator(device=DEV).manual_seed(77)
    for trial in range(12):
        B = int(rng.integers(1, 120))
        ptr = ragged_ptr(rng, B, trial % 4, 700)
        n = int(ptr[-1])
        if n == 0:
            continue
        dtype = torch.bfloat16 if trial % 3 else torch.float16
        x = torch.randn(n, 256, device=DEV, generator=g).to(dtype)
        w = (torch.randn(B, 256, 256, device=DEV, generator=g) / 16).to(dtype)
        bias = torch.randn(B, 256, device=DEV, generator=g).to(dtype) if trial % 2 else None
        try:
            ops.set_matmul_schedule('cyclic')
            ref = ops.segment_matmul(x, ptr, w, bias)
            assert ops.matmul_last_variant().endswith('_k256_wide256r2')
            ops.set_matmul_schedule('auto')
            out = ops.segment_matmul(x, ptr, w, bias)
            assert ops.matmul_last_variant().endswith('_k256_regw')
        finally:
            ops.set_matmul_schedule('auto')
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), (trial, B, n)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_general_shape_kernel_under_noise_is_reproducible(dtype):
    rng = np.random.default_rng(100)
    g = torch.Generator(device=DEV).manual_seed(100)
    shapes = [(int(rng.integers(0, 20000)), int(k), 128) for k in (100, 128, 256, 768, 100, 40, 768, 129)]
    xs = [torch.randn(r, k, device=DEV, generator=g).to(dtype) for r, k, m in shapes]
    ws = [(torch.randn(k, m, device=DEV, generator=g) / k ** 0.5).to(dtype) for r, k, m in shapes]

    def check(outs):
        assert ops.matmul_last_variant().endswith('_gen')
        for a, o, out in zip(xs, ws, outs):
            ref = a.double() @ o.double()
            if dtype == torch.float32:
                assert (out.double() - ref).norm() <= 1e-5 * max(ref.norm().item(), 1e-30)
            else:
                torch.testing.assert_close(out.double(), ref, rtol=2 ** -7, atol=2e-2)

    _repeat_against_first(lambda: ops.grouped_matmul(xs, ws), check)


def test_fused_rgcn_kernel_under_noise_stays_exact():
    """Integer-valued features and signed-permutation weights (every partial sum exact): the fused gather -> MFMA ->
    packed-atomic scatter kernel must give the float64 result in every repetition, whatever order its atomics land."""
    from pyg_lib_amd import rgcn
    g = torch.Generator().manual_seed(4)
    n, F = 5000, 128
    x = torch.randint(-3, 4, (n, F), generator=g).float()
    counts = [30_000, 0, 17, 8192 + 33, 129, 50_000, 1]
    ets = [('a', f'r{i}', 'a') for i in range(len(counts))]
    perm = torch.stack([torch.randperm(F, generator=g) for _ in counts])
    W = torch.zeros(len(counts), F, F)
    W[torch.arange(len(counts))[:, None], perm, torch.arange(F)[None, :]] = \
        (torch.randint(0, 2, (len(counts), F), generator=g) * 2 - 1).float()
    rows, cols = {}, {}
    for et, c in zip(ets, counts):
        rows[et] = torch.sort(torch.randint(0, 2000, (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, n, (c,), generator=g).cuda()
    off = rgcn.type_offsets({'a': n}, ['a'])
    want = torch.zeros(n, F, dtype=torch.float64)
    for i, et in enumerate(ets):
        want.index_add_(0, rows[et].cpu(), x[cols[et].cpu()].double() @ W[i].double())
    assert want.abs().max() <= 256
    xd, Wd = x.bfloat16().cuda(), W.bfloat16().cuda()
    noise = Noise()
    for rep in range(REPS):
        noise.burst()
        y = rgcn.rgcn_layer_fused(xd, off, rows, cols, ets, Wd)
        torch.cuda.synchronize()
        assert torch.equal(y.double().cpu(), want), rep


# ---- kernels that accumulate through hardware float atomics ----------------------------------
# Their results depend on the order the atomics land (fp32 / fp64 / packed 16-bit adds do not commute exactly), so a
# repetition is compared with a float64 reference under the operator's tolerance instead of bit for bit -- except where
# the inputs make every partial sum exact (small integers), where any lost or doubled update, any stale accumulator and
# any write from a neighbour shows as a wrong integer.

def _repeat_exact(make_call, want, reps=REPS, burst=24):
    noise = Noise()
    for rep in range(reps):
        noise.burst(burst)
        out = make_call()
        torch.cuda.synchronize()
        got = out.double().cpu()
        if not torch.equal(got, want):
            bad = (got != want).nonzero()
            raise AssertionError(f'rep {rep}: {bad.size(0)} wrong elements, first at {bad[0].tolist()}: '
                                 f'{got[tuple(bad[0])].item()} != {want[tuple(bad[0])].item()}')


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('K,M', [(64, 64), (128, 128), (256, 256)])
def test_weight_gradient_kernel_under_noise_stays_exact(dtype, K, M):
    """seg_dw_kernel / seg_dw_wide256_kernel / seg_dw_f32_kernel: workgroups add their partial X^T dY slabs into a zeroed
    fp32 accumulator with global_atomic_add_f32, a second kernel rounds.  X in {-1, 0, 1}, dY small integers: every
    partial sum is an exact fp32 integer, so dW must equal the float64 product in every repetition."""
    g = torch.Generator().manual_seed(K + M)
    sizes = [0, 37, 128, 129, 1000, 0, 5000, 31, 257, 9000]
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n, B = int(ptr[-1]), len(sizes)
    # sparse X keeps the 9000-row relation's sums small: |dW| < 256 is exact in bf16 too
    x = torch.randint(-1, 2, (n, K), generator=g).float() * (torch.rand(n, K, generator=g) < 0.06)
    gy = torch.randint(-2, 3, (n, M), generator=g).float()
    want = torch.stack([x[ptr[b]:ptr[b + 1]].double().t() @ gy[ptr[b]:ptr[b + 1]].double() for b in range(B)])
    assert want.abs().max() < 256   # exact in bf16 as well
    xd = x.to(dtype).to(DEV)
    gyd = gy.to(dtype).to(DEV)
    wd = torch.zeros(B, K, M, dtype=dtype, device=DEV, requires_grad=True)

    def call():
        y = ops.segment_matmul(xd, ptr if call.host else ptr.to(DEV), wd)
        call.host = not call.host
        return torch.autograd.grad(y, [wd], gyd)[0]
    call.host = True
    _repeat_exact(call, want)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16, torch.float64])
@pytest.mark.parametrize('K', [128, 5, 96])
@pytest.mark.parametrize('sorted_index', [False, True])
def test_scatter_sum_under_noise_stays_exact(dtype, K, sorted_index):
    """scatter_sum_vec_kernel (16-byte rows, native fp32 / fp64 / packed bf16 / f16 atomics), scatter_elem_kernel (K = 5)
    and the sorted-run path of segment_sum_coo: integer-valued sources whose bucket sums stay below 256."""
    g = torch.Generator().manual_seed(K)
    E, N = 20000, 700
    src = torch.randint(-2, 3, (E, K), generator=g).float()
    index = torch.randint(0, N, (E,), generator=g)
    if sorted_index:
        index = index.sort().values
    want = torch.zeros(N, K, dtype=torch.float64).index_add_(0, index, src.double())
    assert want.abs().max() < 256
    sd, idx = src.to(dtype).to(DEV), index.to(DEV)
    if sorted_index:
        _repeat_exact(lambda: ops.segment_sum_coo(sd, idx, dim_size=N), want)
    else:
        _repeat_exact(lambda: ops.scatter_sum(sd, idx, 0, None, N), want)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_scatter_sum_into_a_caller_buffer_under_noise_stays_exact(dtype):
    """`out=` accumulation (no zero fill by the operator) and a fresh output allocated right after a large block was
    returned to the caching allocator: the accumulator's initial contents are exactly what the caller / the operator
    wrote, never what the block held before."""
    g = torch.Generator().manual_seed(9)
    E, N, K = 30000, 1500, 128
    src = torch.randint(-2, 3, (E, K), generator=g).float()
    index = torch.randint(0, N, (E,), generator=g)
    base = torch.randint(-4, 5, (N, K), generator=g).float()
    want_out = base.double().index_add(0, index, src.double())
    want_new = torch.zeros(N, K, dtype=torch.float64).index_add_(0, index, src.double())
    sd, idx = src.to(dtype).to(DEV), index.to(DEV)

    def into_out():
        out = base.to(dtype).to(DEV)
        return ops.scatter_sum(sd, idx, 0, out, N)

    def after_free():
        junk = torch.full((N, K), 77.0, dtype=dtype, device=DEV)   # a block of the output's size, dirty, freed ...
        del junk
        return ops.scatter_sum(sd, idx, 0, None, N)               # ... and most likely handed out again here

    _repeat_exact(into_out, want_out, reps=16)
    _repeat_exact(after_free, want_new, reps=16)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_segment_sum_csr_under_noise_stays_exact(dtype):
    from pyg_lib_amd import ops as o
    g = torch.Generator().manual_seed(10)
    N, K = 4000, 128
    deg = torch.randint(0, 12, (N,), generator=g)
    deg[5] = 3000
    indptr = torch.cat([torch.zeros(1, dtype=torch.long), deg.cumsum(0)])
    E = int(indptr[-1])
    src = torch.randint(-2, 3, (E, K), generator=g).float()
    seg = torch.repeat_interleave(torch.arange(N), deg)
    want = torch.zeros(N, K, dtype=torch.float64).index_add_(0, seg, src.double())
    assert want.abs().max() < 256   # one 3000-element row included: fp32 partial sums, one rounding
    sd, ip = src.to(dtype).to(DEV), indptr.to(DEV)
    _repeat_exact(lambda: o.segment_sum_csr(sd, ip), want, reps=16)


def test_bench_two_ranks_on_one_device_exercises_the_sharded_branch():
    """`bench.py --gpus 2` with both ranks on cuda:0 over gloo (RCCL refuses two ranks per device): the N > 1 branch
    -- row shards, barrier + max-over-ranks timing, the all-gather leg, C4's LPT shards with the in-place gather --
    runs on every driver pass, and its JSON line has the shape the driver reads."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # no launcher: bench.py re-executes itself under torch.distributed.run with two ranks (what `python bench.py --gpus N`
    # does on a multi-GPU node with the nccl backend)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3',
           '--warmup', '1', '--scale', '0.05', '--debug-one-device', '--no-sampler', '--no-cpu-baseline']
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    r = json.loads(line)
    assert r['n_gpus'] == 2 and r['scaling'] == 'strong' and r['value'] > 0 and r['steps'] == 3
    assert r['config']['rows_per_gpu'] * 2 <= r['config']['rows'] + 1
    assert 'ms' in r['allgather'] and r['allgather']['backend'] == 'gloo', r['allgather']
    assert r['c4']['n_gpus'] == 2 and 'incl_allgather' in r['c4'] and r['c4']['compute_only']['ms'] > 0, r['c4']
    assert r['roofline']['kernel'].startswith('mfma_')


# ---- the compare-and-swap flavour of the remaining float atomics, the in-process self-test --------

@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16, torch.float64])
@pytest.mark.parametrize('K', [128, 5])
def test_scatter_sum_cas_mode_is_exact(dtype, K):
    """`PYG_HIP_FLOAT_ATOMICS=cas` / pyg_hip_set_float_atomic_mode(1): scatter_sum_vec_kernel (16-byte rows) and
    scatter_elem_kernel (K = 5, float64) add through compare-and-swap loops on the containing word instead of the
    hardware's floating-point atomic unit.  Integer-valued data: both flavours must give the float64 sums exactly."""
    from pyg_lib_amd import diagnostics
    g = torch.Generator().manual_seed(K + 1)
    E, N = 20000, 700
    src = torch.randint(-2, 3, (E, K), generator=g).float()
    index = torch.randint(0, N, (E,), generator=g)
    want = torch.zeros(N, K, dtype=torch.float64).index_add_(0, index, src.double())
    sd, idx = src.to(dtype).to(DEV), index.to(DEV)
    before = diagnostics.set_float_atomic_mode('cas')
    try:
        _repeat_exact(lambda: ops.scatter_sum(sd, idx, 0, None, N), want, reps=6)
        assert 'compare-and-swap' in diagnostics.last_accumulate_info()
        base = torch.randint(-4, 5, (N, K), generator=g).float()
        out = ops.scatter_sum(sd, idx, 0, base.to(dtype).to(DEV), N)
        assert torch.equal(out.double().cpu(), base.double() + want)
    finally:
        assert diagnostics.set_float_atomic_mode(before) == 'cas'
    _repeat_exact(lambda: ops.scatter_sum(sd, idx, 0, None, N), want, reps=2)
    assert 'hardware' in diagnostics.last_accumulate_info()


def test_fused_rgcn_cas_mode_is_exact():
    from pyg_lib_amd import diagnostics, rgcn
    g = torch.Generator().manual_seed(5)
    n, F = 3000, 128
    x = torch.randint(-3, 4, (n, F), generator=g).float()
    counts = [20_000, 0, 17, 4096 + 33]
    ets = [('a', f'r{i}', 'a') for i in range(len(counts))]
    perm = torch.stack([torch.randperm(F, generator=g) for _ in counts])
    W = torch.zeros(len(counts), F, F)
    W[torch.arange(len(counts))[:, None], perm, torch.arange(F)[None, :]] = \
        (torch.randint(0, 2, (len(counts), F), generator=g) * 2 - 1).float()
    rows, cols = {}, {}
    for et, c in zip(ets, counts):
        rows[et] = torch.sort(torch.randint(0, 1500, (c,), generator=g)).values.cuda()
        cols[et] = torch.randint(0, n, (c,), generator=g).cuda()
    off = rgcn.type_offsets({'a': n}, ['a'])
    want = torch.zeros(n, F, dtype=torch.float64)
    for i, et in enumerate(ets):
        want.index_add_(0, rows[et].cpu(), x[cols[et].cpu()].double() @ W[i].double())
    assert want.abs().max() <= 256
    for dtype in (torch.bfloat16, torch.float16):
        xd, Wd = x.to(dtype).cuda(), W.to(dtype).cuda()
        before = diagnostics.set_float_atomic_mode('cas')
        try:
            for rep in range(4):
                y = rgcn.rgcn_layer_fused(xd, off, rows, cols, ets, Wd)
                assert torch.equal(y.double().cpu(), want), (dtype, rep)
            assert 'pyg_hip_rgcn_fused' in diagnostics.last_accumulate_info() and 'compare-and-swap' in diagnostics.last_accumulate_info()
        finally:
            diagnostics.set_float_atomic_mode(before)
        y = rgcn.rgcn_layer_fused(xd, off, rows, cols, ets, Wd)
        assert torch.equal(y.double().cpu(), want)


def test_atomic_selftest_is_clean_and_reports_the_memory_it_ran_on():
    """pyg_hip_atomic_selftest on a block of the caching allocator and torch's current stream: 30 variants (5 add flavours x
    3 ways of clearing x 2 readbacks), every one must deliver every update -- also on a side stream with a noise stream
    next to it."""
    from pyg_lib_amd import diagnostics
    bad, text = diagnostics.atomic_selftest(rounds=3)
    assert bad == 0, text
    assert 'device memory' in text and '0 of 30 variants bad' in text
    noise = Noise()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        noise.burst(8)
        bad, text = diagnostics.atomic_selftest(rounds=2, megabytes=4)
    assert bad == 0, text
    torch.cuda.synchronize()
