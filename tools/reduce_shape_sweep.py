"""scatter / segment / gather ops over row widths K and dtypes: ms and fraction of 8 TB/s on the algorithmic bytes, next to torch's
own op where it has one.  python tools/reduce_shape_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
N = 1_000_000
for dtype in (torch.float32, torch.bfloat16):
    esz = torch.empty(0, dtype=dtype).element_size()
    for K in (1, 4, 16, 64, 128, 256):
        E = max(1_000_000, min(20_000_000, 2_000_000_000 // (K * esz)))
        src = torch.randn(E, K, device=dev, generator=g).to(dtype)
        idx = torch.randint(0, N, (E,), device=dev, generator=g)
        sidx = torch.sort(idx).values
        alg = esz * (E * K + N * K) + 8 * E
        rows = []
        for name, fn in (('scatter_sum', lambda: ops.scatter_sum(src, idx, 0, None, N)),
                         ('scatter_max', lambda: ops.scatter_max(src, idx, 0, None, N)),
                         ('segment_sum_coo', lambda: ops.segment_sum_coo(src, sidx, None, N)),
                         ('gather_coo', lambda: ops.gather_coo(src[:N], sidx)),
                         ('torch.index_add_', lambda: torch.zeros(N, K, device=dev, dtype=dtype).index_add_(0, idx, src))):
            ms = bench_legs._event_ms(fn, 3, warmup=1)
            rows.append(f'{name} {ms:.3f} ms ({alg / ms / 8e9:.2f})')
        print(f'{str(dtype)[6:]:9s} K={K:3d} E={E:9d}: ' + ' | '.join(rows), flush=True)
        del src, idx, sidx
