"""Which operators have a cliff when ONE destination collects a share of the positions: ms without a hub | with a hub of 2.5 % |
of 25 % of 8 M positions, per operator and row width.   python tools/hub_sweep2.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
E, N = 8_000_000, 1_000_000
for dtype, K in ((torch.float32, 1), (torch.float32, 4), (torch.float32, 16), (torch.bfloat16, 16), (torch.float32, 64), (torch.bfloat16, 128)):
    src = torch.randn(E, K, device=dev, generator=g).to(dtype)
    rows = {}
    for hub in (0, 200_000, 2_000_000):
        idx = torch.randint(0, N, (E,), device=dev, generator=g)
        if hub:
            idx[torch.randperm(E, device=dev, generator=g)[:hub]] = 12345
        sidx = torch.sort(idx).values
        ptr = torch.zeros(N + 1, dtype=torch.long, device=dev)
        ptr[1:] = torch.bincount(sidx, minlength=N).cumsum(0)
        small = src[:N]
        legs = {
            'scatter_sum': lambda: ops.scatter_sum(src, idx, 0, None, N),
            'scatter_max': lambda: ops.scatter_max(src, idx, 0, None, N),
            'scatter_mean': lambda: ops.scatter_mean(src, idx, 0, None, N),
            'segment_sum_coo': lambda: ops.segment_sum_coo(src, sidx, None, N),
            'segment_max_coo': lambda: ops.segment_max_coo(src, sidx, None, N),
            'gather_coo': lambda: ops.gather_coo(small, sidx),
            'segment_sum_csr': lambda: ops.segment_sum_csr(src, ptr),
            'segment_max_csr': lambda: ops.segment_max_csr(src, ptr),
            'gather_csr': lambda: ops.gather_csr(small, ptr),
            'torch.index_add_': lambda: torch.zeros(N, K, device=dev, dtype=dtype).index_add_(0, idx, src),
        }
        if dtype == torch.float32:
            legs['softmax_csr'] = lambda: ops.softmax_csr(src, ptr)
            legs['scatter_softmax'] = lambda: ops.scatter_softmax(src, idx, 0, N)
        for name, fn in legs.items():
            rows.setdefault(name, []).append(bench_legs._event_ms(fn, 2, warmup=1))
    for name, v in rows.items():
        flag = '   <-- cliff' if v[2] > 4 * v[0] + 0.2 else ''
        print(f'{str(dtype)[6:]:9s} K={K:3d} {name:18s} {v[0]:9.3f} | {v[1]:9.3f} | {v[2]:9.3f} ms{flag}', flush=True)
