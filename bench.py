"""bench.py -- headline measurement of the hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input, inputs resident in HBM:
``segment_matmul`` on BASELINE.json configs[1] (154 relations, 21,111,007 rows, F=128, bf16).
`value` is whole-job GFLOP/s (2*N*K*M flop per step / wall time, max over ranks).

N > 1 ranks: STRONG scaling -- the same fixed C2 job, its relation list cut at row boundaries into N contiguous
shards (pyg_lib_amd/sharding.py), every rank multiplies its shard, outputs left sharded: the timed region has no
data-path collective (north_star's ">= 6x at 8 GPUs" refers to this compute-only figure; SURVEY.md 8(e) shows the
all-gather of the outputs is xGMI-bound and ~26x slower than the HBM-bound shard compute).  The all-gather is timed
on its own and reported under "allgather"; BASELINE config C4 (grouped_matmul, 512 groups, relation-sharded by LPT,
in-place RCCL all-gather) is the "c4" object with `compute_only` and `incl_allgather` side by side.
(``--scaling weak`` grows the job with N instead: debugging only.)

The JSON line also carries
  roofline      dominant kernel vs the HBM roofline: algorithmic bytes / HIP-event kernel time on the launch stream
                (pyg_hip_profile_*), plus `achievable`: what hand-written device copies of the same 1 read : 1 write
                byte mix reach on this box on the same two buffers (pyg_hip_stream_copy: fine-grained sweep = the best
                copy known on this hardware; contiguous / cyclic = the static tile schedules of the two older kernels
                without the arithmetic -- the default ticket-schedule kernel draws its tiles in address order and is
                not tied to either)
  cpu_baseline  the reference's CPU arithmetic -- one at::matmul per relation (ops/cpu/matmul_kernel.cpp:195-201)
                -- as per-segment torch.matmul on CPU tensors of the same dtype, all host threads; `cpu_port` is the
                oracle's own C restatement (test infrastructure) on a smaller sample
  sampler       neighbor_sample sampled-edges/s on the C3-shaped synthetic graph (single GPU by design)
  segment_matmul_f32   C2 in fp32 (north_star's 1e-5 parity configuration): split-bf16 MFMAs, bound = HBM; `exact` =
                       the fp32 MFMA kernel (bound = fp32 MFMA rate)
  segment_short        4 Mi rows in 16 384 relations of 256 rows (the item-ring kernel; the ticket kernel beside it)
  grouped_mixed        per-group shapes (K in {100, 128, 256, 768}) through the general-shape MFMA kernel + the
                       one-thread-per-output kernel's time beside it; segment_matmul with K = 100
  c4, c5, index_sort, scatter_sum, segment_matmul_backward   the other configs / ops (bench_legs.py)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
F32_MFMA_PEAK_TFLOPS = 157.0  # dense v_mfma_f32_32x32x2_f32 peak at 2.4 GHz (MI355X_MICROARCH.md)

C2 = dict(B=154, N=21_111_007, F=128)


def c2_ptr(N, B):
    g = torch.Generator(device='cpu').manual_seed(0)
    frac = torch.rand(B, generator=g)
    sizes = torch.floor(frac / frac.sum() * N).long()
    sizes[-1] += N - sizes.sum()
    return torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])


def make_c2(device, rank, world, dtype=torch.bfloat16, scale=1.0):
    B, N, F = C2['B'], int(C2['N'] * scale), C2['F']
    ptr = c2_ptr(N, B)
    # contiguous row shard of this rank, ptr clipped to it (pyg_lib_amd/sharding.py)
    from pyg_lib_amd import sharding
    r0, r1, lptr = sharding.shard_ptr(ptr, rank, world)
    gd = torch.Generator(device=device).manual_seed(1 + rank)
    x = torch.empty(r1 - r0, F, device=device, dtype=dtype)
    step = 4_000_000
    for s in range(0, r1 - r0, step):
        e = min(s + step, r1 - r0)
        x[s:e] = torch.randn(e - s, F, device=device, generator=gd, dtype=torch.float32).to(dtype)
    w = (torch.randn(B, F, F, device=device, generator=gd, dtype=torch.float32) / F ** 0.5).to(dtype)
    return x, lptr, w, (N, B, F)


def cpu_baseline_aten(dtype='bf16', rows=None, budget_s=25.0):
    """The reference's CPU path for this op is a loop of at::matmul_out over the relations
    (ops/cpu/matmul_kernel.cpp:195-201, 428-434): the same arithmetic as per-segment torch.matmul on CPU tensors.
    Sample: the FULL C2 relation list (the GPU run's own ptr: 154 relations, 21,111,007 rows), all host threads, best of
    up to 6 passes within the time budget.  The rows are a 1 Mi-row random block repeated (GEMM time does not depend
    on the values; drawing 2.7e9 normals on the host would take longer than the measurement)."""
    B, F = C2['B'], C2['F']
    rows = C2['N'] if rows is None else rows
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    ptr = c2_ptr(rows, B).tolist()
    g = torch.Generator().manual_seed(0)
    blk = torch.randn(1 << 20, F, generator=g).to(tdt)
    x = blk.repeat((rows + blk.size(0) - 1) // blk.size(0), 1)[:rows].contiguous()
    w = (torch.randn(B, F, F, generator=g) / F ** 0.5).to(tdt)
    out = torch.empty(rows, F, dtype=tdt)

    def run():
        for b in range(B):
            if ptr[b + 1] > ptr[b]:
                torch.matmul(x[ptr[b]:ptr[b + 1]], w[b], out=out[ptr[b]:ptr[b + 1]])

    run()
    best, reps, t_all = None, 0, time.perf_counter()
    while reps < 6 and time.perf_counter() - t_all < budget_s:
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        reps += 1
    return dict(value=round(2.0 * rows * F * F / best / 1e9, 2), unit='GFLOP/s', cores=torch.get_num_threads(),
                kind='reference', via='aten-per-segment',
                sample=f'per-relation torch.matmul on CPU {dtype} tensors (= at::matmul_out per segment, '
                       f'ops/cpu/matmul_kernel.cpp:195-201): the full C2 relation list ({B} relations, {rows} rows, F=128; a 1 Mi-row '
                       f'random block repeated), best of {reps} passes, {torch.get_num_threads()} threads')


def cpu_port_segment_matmul(sample_rows=480_000, dtype='bf16'):
    """The oracle's C restatement (kind "port", test infrastructure) on a bounded sample."""
    import oracle
    B, F = 8, C2['F']
    rng = np.random.default_rng(0)
    sizes = np.full(B, sample_rows // B)
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(ptr[-1])
    x = rng.standard_normal((n, F), dtype=np.float32)
    w = rng.standard_normal((B, F, F), dtype=np.float32) / F ** 0.5
    code = oracle.F32
    if dtype == 'bf16':
        x, w, code = oracle.f32_to_bf16_bits(x), oracle.f32_to_bf16_bits(w), oracle.BF16
    oracle.segment_matmul(x[:1024], np.array([0, 1024]), w[:1], dtype=code)  # warm up / build
    best, reps, t_all = None, 0, time.perf_counter()
    while reps < 10 and time.perf_counter() - t_all < 5.0:
        t0 = time.perf_counter()
        oracle.segment_matmul(x, ptr, w, dtype=code)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        reps += 1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    threads = int(os.environ.get('OMP_NUM_THREADS', cores))
    return dict(value=round(2.0 * n * F * F / best / 1e9, 2), unit='GFLOP/s', cores=threads, kind='port',
                sample=f'oracle/oracle_matmul.c segment_matmul, {B} relations x {sample_rows // B} rows, '
                       f'F=128 {dtype}, best of {reps}, OpenMP')


def stream_copy_rates(L, x, reps=5):
    """Hand-written copies of x into an out-sized buffer (1 read : 1 write, the kernel's byte mix)."""
    dst = torch.empty_like(x)
    nbytes = x.numel() * x.element_size()
    stream = torch.cuda.current_stream().cuda_stream
    rates = {}
    for name, mode in (('fine_sweep', 0), ('contiguous_ranges', 1), ('cyclic', 2)):
        for _ in range(2):
            L.pyg_hip_stream_copy(x.data_ptr(), dst.data_ptr(), nbytes, mode, stream)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.pyg_hip_stream_copy(x.data_ptr(), dst.data_ptr(), nbytes, mode, stream)
        e1.record()
        torch.cuda.synchronize()
        rates[name] = round(2.0 * nbytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9, 1)
    ok = bool(torch.equal(dst.view(torch.int16)[-4096:], x.view(torch.int16)[-4096:]))
    del dst
    return rates, ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--scale', type=float, default=1.0, help='shrink the workload (debug only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sampler', action='store_true')
    ap.add_argument('--no-legs', action='store_true', help='skip the c4 / c5 / index_sort / scatter / backward legs')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--scaling', default='strong', choices=['strong', 'weak'])
    ap.add_argument('--schedule', default='auto', choices=['auto', 'contiguous', 'cyclic', 'ticket'],
                    help='tile schedule of the bf16 kernel (PYG_HIP_MM_SCHED_* of pyg_hip.h)')
    ap.add_argument('--debug-one-device', action='store_true',
                    help='debug only: all ranks share cuda:0 over gloo (exercises the N>1 code path on a 1-GPU box)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become N ranks, one per GPU (what the driver's own
        # `python -m torch.distributed.run ... bench.py --gpus N` command does)
        have = torch.cuda.device_count()
        if not args.debug_one_device and have < args.gpus:
            sys.exit(f'bench.py: --gpus {args.gpus} asked for, but only {have} HIP device(s) are visible')
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execvpe(sys.executable, cmd, env)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    import torch.distributed as dist
    distributed = world > 1
    if args.debug_one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    backend = None
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = 'gloo' if args.debug_one_device else 'nccl'
        if args.debug_one_device:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)

    import pyg_lib_amd
    from pyg_lib_amd import ops, _capi
    L = _capi.lib()
    L.pyg_hip_profile_enable.argtypes = [ctypes.c_int]
    L.pyg_hip_profile_collect.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.pyg_hip_profile_collect.restype = ctypes.c_int
    L.pyg_hip_stream_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    L.pyg_hip_stream_copy.restype = ctypes.c_int
    ops.set_matmul_schedule(args.schedule)

    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    job_scale = args.scale * (world if args.scaling == 'weak' else 1)
    x, ptr, w, (N, B, F) = make_c2(device, rank, world, dtype, job_scale)
    esz = x.element_size()

    def step():
        return ops.segment_matmul(x, ptr, w)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    L.pyg_hip_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    buf = (ctypes.c_float * max(args.steps, 1))()
    nk = L.pyg_hip_profile_collect(buf, args.steps)
    L.pyg_hip_profile_enable(0)
    kernel_ms = float(np.mean([buf[i] for i in range(min(nk, args.steps))])) if nk else float('nan')
    variant = ops.matmul_last_variant()

    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    flops = 2.0 * N * F * F
    value = flops / (elapsed / args.steps) / 1e9

    # dominant kernel vs HBM roofline (this rank's shard): algorithmic bytes per launch
    n_local = x.size(0)
    alg_bytes = esz * (n_local * F + n_local * F + B * F * F) + 8 * (B + 1)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms == kernel_ms else None
    # HBM bytes from PMC (FETCH_SIZE / WRITE_SIZE, separate --pmc passes of this command, corrected as
    # MI355X_MICROARCH.md prescribes): recorded in profiles/ per kernel variant, valid for the full single-GPU launch
    traffic = None
    traffic_source = None
    if world == 1 and args.scale == 1.0 and args.dtype == 'bf16':
        for name in ('r6_segment_matmul_c2_pmc.json', 'r5_segment_matmul_c2_pmc.json', 'r4_segment_matmul_c2_pmc.json', 'r3_segment_matmul_c2_pmc.json', 'r2_segment_matmul_c2_pmc.json',
                     'r1_segment_matmul_c2_pmc.json'):
            pmc = os.path.join(ROOT, 'profiles', name)
            if not os.path.exists(pmc):
                continue
            try:
                rec = json.load(open(pmc))
                if rec.get('kernel_variant', 'mfma_bf16_k128_mc128') == variant:
                    traffic = int(rec['hbm_traffic_bytes'])
                    traffic_source = 'profiles/' + name + ' (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)'
                    break
            except Exception:  # noqa: BLE001
                pass
    tflops = None if achieved is None else 2.0 * n_local * F * F / (kernel_ms * 1e-3) / 1e12
    if args.dtype == 'f32':
        # fp32, F = 128: AI = 32 flop/B is above the fp32 ridge (157 TF / 8 TB/s = 20): bound by the fp32 MFMA rate
        roofline = dict(bound='mfma', achieved=None if tflops is None else round(tflops, 1), peak=F32_MFMA_PEAK_TFLOPS,
                        unit='TFLOP/s', frac=None if tflops is None else round(tflops / F32_MFMA_PEAK_TFLOPS, 4),
                        traffic=traffic, traffic_source=traffic_source, kernel=variant, kernel_ms=round(kernel_ms, 4),
                        alg_bytes=int(alg_bytes), hbm_GBps=None if achieved is None else round(achieved, 1))
    else:
        roofline = dict(bound='hbm', achieved=None if achieved is None else round(achieved, 1), peak=HBM_PEAK_GBS,
                        unit='GB/s', frac=None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                        traffic=traffic, traffic_source=traffic_source, kernel=variant, kernel_ms=round(kernel_ms, 4),
                        alg_bytes=int(alg_bytes), mfma_tflops=None if tflops is None else round(tflops, 1))

    # context for `frac`: hand-written copies of the same byte mix on this box, right after the timed region
    if rank == 0 and world == 1 and args.dtype == 'bf16':
        try:
            rates, ok = stream_copy_rates(L, x)
            roofline['achievable'] = dict(GBps=max(rates.values()), copies_GBps=rates, verified=ok,
                                          what='pyg_hip_stream_copy of x into an out-sized buffer (1 read : 1 write)')
            if achieved is not None:
                roofline['frac_of_achievable'] = round(achieved / max(rates.values()), 4)
        except Exception as e:  # noqa: BLE001 - context only
            roofline['achievable'] = dict(error=repr(e)[:200])

    allgather = None
    if distributed:
        # RCCL all-gather(v) of the sharded outputs over xGMI, timed on its own (see module docstring)
        try:
            from pyg_lib_amd import sharding
            sharding.all_gather_rows(out, N)  # warm-up
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            reps = 3
            for _ in range(reps):
                full = sharding.all_gather_rows(out, N)
            torch.cuda.synchronize()
            dist.barrier()
            tg = torch.tensor([(time.perf_counter() - ta) / reps], device=device, dtype=torch.float64)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            ag_ms = float(tg.item()) * 1e3
            allgather = dict(ms=round(ag_ms, 3), bytes_per_rank_received=int((N - n_local) * F * esz),
                             value_incl_allgather=round(flops / ((ms_per_step + ag_ms) * 1e-3) / 1e9, 1), backend=backend)
            del full
        except Exception as e:  # noqa: BLE001 - the headline line must survive a collective failure
            allgather = dict(error=repr(e)[:200])

    result = None
    if rank == 0:
        result = {
            'metric': 'segment_matmul GFLOP/s + sampled-edges/sec, 1/2/4/8 MI355X vs CPU ref',
            'value': round(value, 1), 'unit': 'GFLOP/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
            'scaling': args.scaling, 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'segment_matmul ogbn-mag-shaped: 154 relations, 21,111,007 rows, '
                                   'F_in=F_out=128 (BASELINE.json configs[1])',
                       'relations': B, 'rows': N, 'rows_per_gpu': N // world, 'F': F, 'scale': args.scale,
                       'sharding': f'contiguous row shards x{world}, outputs left sharded (compute only)'},
            'roofline': roofline,
        }
        if allgather is not None:
            result['allgather'] = allgather

    del out
    # C4 (grouped_matmul, relation-sharded): runs on every rank
    if not args.no_legs and args.dtype == 'bf16':
        import bench_legs
        try:
            c4 = bench_legs.leg_c4(device, rank, world)
        except Exception as e:  # noqa: BLE001
            c4 = dict(error=repr(e)[:300])
        if rank == 0:
            result['c4'] = c4
    if rank == 0 and world == 1 and not args.no_legs and args.dtype == 'bf16':
        import bench_legs
        for key, fn in (('segment_matmul_backward', lambda: bench_legs.leg_backward(device, x, ptr, w)),):
            try:
                result[key] = fn()
            except Exception as e:  # noqa: BLE001
                result[key] = dict(error=repr(e)[:300])
    del x
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_legs and args.dtype == 'bf16':
        for key, fn in (('segment_matmul_f32', lambda: bench_legs.leg_segment_matmul_f32(device, make_c2)),
                        ('grouped_mixed', lambda: bench_legs.leg_grouped_mixed(device)),
                        ('segment_short', lambda: bench_legs.leg_segment_short(device)),
                        ('c5', lambda: bench_legs.leg_c5(device)),
                        ('index_sort', lambda: bench_legs.leg_index_sort(device)),
                        ('scatter_sum', lambda: bench_legs.leg_scatter_sum(device))):
            try:
                result[key] = fn()
            except Exception as e:  # noqa: BLE001
                result[key] = dict(error=repr(e)[:300])
            torch.cuda.empty_cache()
    # secondary metric: sampled-edges/s of the HIP neighbour sampler (single GPU by design)
    if rank == 0 and not args.no_sampler:
        try:
            import bench_sampler as sampler_bench
            result['sampler'] = sampler_bench.run(device)
        except ImportError:
            result['sampler'] = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline_aten(dtype=args.dtype)
        try:
            result['cpu_port'] = cpu_port_segment_matmul(dtype=args.dtype)
        except Exception as e:  # noqa: BLE001
            result['cpu_port'] = dict(error=repr(e)[:200])
    elif rank == 0:
        result['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
