"""Kernel-time survey of the general-shape MFMA matmul kernel (matmul_gen.hip): HIP events around the kernel
(pyg_hip_profile_*), algorithmic bytes s*(N*K + N*M + K*M) per group, fraction of the 8 TB/s HBM peak.

    python tools/bench_gen.py [case ...]
"""
import ctypes
import math
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from pyg_lib_amd import ops, _capi  # noqa: E402

L = _capi.lib()
L.pyg_hip_profile_enable.argtypes = [ctypes.c_int]
L.pyg_hip_profile_collect.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.pyg_hip_profile_collect.restype = ctypes.c_int
DEV = 'cuda:0'


def kernel_ms(fn, iters=10, warmup=3):
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
    v = [buf[i] for i in range(min(n, iters))]
    return float(np.mean(v)), float(np.min(v))


def report(name, alg_bytes, flops, fn):
    mean, best = kernel_ms(fn)
    print(f'{name:44s} {ops.matmul_last_variant():26s} {mean:8.3f} ms (min {best:.3f})  '
          f'{alg_bytes / mean / 1e6:8.1f} GB/s  frac {alg_bytes / mean / 1e6 / 8000:.3f}  {flops / mean / 1e9:7.1f} TF',
          flush=True)


def seg_case(N, K, M, B, dtype, sched=None):
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(N, K, device=DEV, generator=g).to(dtype)
    w = (torch.randn(B, K, M, device=DEV, generator=g) / K ** 0.5).to(dtype)
    fr = torch.rand(B)
    sizes = torch.floor(fr / fr.sum() * N).long()
    sizes[-1] += N - sizes.sum()
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    s = x.element_size()
    if sched:
        ops.set_matmul_schedule(sched)
    report(f'segment N={N} K={K} M={M} B={B} {str(dtype)[6:]} {sched or ""}', s * (N * K + N * M + B * K * M), 2.0 * N * K * M,
           lambda: ops.segment_matmul(x, ptr, w))
    ops.set_matmul_schedule('auto')


def grouped_mixed(dtype, rows_total=6_000_000, G=64, ks=(100, 128, 256, 768), M=128, sched=None):
    g = torch.Generator().manual_seed(0)
    rows = torch.exp(torch.rand(G, generator=g) * (math.log(65536.0 * 4) - math.log(1024.0)) + math.log(1024.0))
    rows = (rows / rows.sum() * rows_total).long().tolist()
    gd = torch.Generator(device=DEV).manual_seed(1)
    xs = [torch.randn(r, ks[i % len(ks)], device=DEV, generator=gd).to(dtype) for i, r in enumerate(rows)]
    ws = [(torch.randn(ks[i % len(ks)], M, device=DEV, generator=gd) / 16).to(dtype) for i in range(G)]
    s = xs[0].element_size()
    alg = sum(s * (r * ks[i % len(ks)] + r * M + ks[i % len(ks)] * M) for i, r in enumerate(rows))
    fl = sum(2.0 * r * ks[i % len(ks)] * M for i, r in enumerate(rows))
    if sched:
        ops.set_matmul_schedule(sched)
    report(f'grouped mixed K={ks} M={M} G={G} {str(dtype)[6:]} {sched or ""}', alg, fl, lambda: ops.grouped_matmul(xs, ws))
    ops.set_matmul_schedule('auto')


CASES = {
    'k100': lambda: seg_case(8_000_000, 100, 128, 47, torch.bfloat16),
    'k100f': lambda: seg_case(4_000_000, 100, 128, 47, torch.float32),
    'k128': lambda: seg_case(8_000_000, 128, 128, 154, torch.bfloat16),
    'k128gen': lambda: seg_case(8_000_000, 128, 128, 154, torch.bfloat16, 'general'),
    'k128genf': lambda: seg_case(4_000_000, 128, 128, 154, torch.float32, 'general'),
    'k128f': lambda: seg_case(4_000_000, 128, 128, 154, torch.float32),
    'k256gen': lambda: seg_case(4_000_000, 256, 256, 154, torch.bfloat16, 'general'),
    'k768': lambda: seg_case(2_000_000, 768, 128, 16, torch.bfloat16),
    'k64m64': lambda: seg_case(8_000_000, 64, 64, 154, torch.bfloat16, 'general'),
    'mixed': lambda: grouped_mixed(torch.bfloat16),
    'mixedf': lambda: grouped_mixed(torch.float32, rows_total=3_000_000),
    'mixed_naive': lambda: grouped_mixed(torch.bfloat16, sched='naive'),
}

if __name__ == '__main__':
    for c in (sys.argv[1:] or list(CASES)):
        CASES[c]()
