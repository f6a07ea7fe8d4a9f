"""scatter_sum / segment_sum_coo with one hub destination (a fraction of all edges point at one row): the row kernels give a row to
one 16-lane group.   python tools/hub_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
E, N, K = 8_000_000, 1_000_000, 128
src = torch.randn(E, K, device=dev, generator=g).bfloat16()
for hub in (0, 1000, 20_000, 200_000, 2_000_000):
    idx = torch.randint(0, N, (E,), device=dev, generator=g)
    if hub:
        idx[torch.randperm(E, device=dev, generator=g)[:hub]] = 12345
    sidx = torch.sort(idx).values
    a = bench_legs._event_ms(lambda: ops.scatter_sum(src, idx, 0, None, N), 3, warmup=1)
    b = bench_legs._event_ms(lambda: ops.segment_sum_coo(src, sidx, None, N), 3, warmup=1)
    t = bench_legs._event_ms(lambda: torch.zeros(N, K, device=dev, dtype=torch.bfloat16).index_add_(0, idx, src), 3, warmup=1)
    print(f'hub of {hub:8d} edges: scatter_sum {a:8.3f} ms | segment_sum_coo {b:8.3f} ms | torch.index_add_ {t:8.3f} ms', flush=True)
