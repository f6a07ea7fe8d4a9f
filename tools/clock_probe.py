"""Shader clock under the matmul kernels (pyg_hip_clock_probe): idle, C2 bf16 (HBM-bound), C2 fp32 exact (MFMA-bound),
C2 fp32 split-bf16.  python tools/clock_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
print('idle:', round(bench_legs._clock_under(lambda: None, 1.0, dev, 20.0)), 'MHz')
for dtype, name in ((torch.bfloat16, 'bf16'), (torch.float32, 'fp32')):
    x, ptr, w, (N, B, F) = bench.make_c2(dev, 0, 1, dtype, 1.0)
    flop = 2.0 * N * F * F
    modes = [None] if dtype == torch.bfloat16 else [False, True]
    for split in modes:
        ctx = ops.matmul_f32_split(split) if split is not None else None
        if ctx:
            ctx.__enter__()
        ms = bench_legs._kernel_ms(lambda: ops.segment_matmul(x, ptr, w), iters=5, warmup=2)
        mhz = bench_legs._clock_under(lambda: ops.segment_matmul(x, ptr, w), ms, dev)
        if ctx:
            ctx.__exit__(None, None, None)
        print(f'C2 {name}{"" if split is None else (" split" if split else " exact")}: {ms:.3f} ms, {flop / ms / 1e9:.1f} TFLOP/s, '
              f'{mhz:.0f} MHz under load ({ops.matmul_last_variant()})')
    del x, w
