"""index_sort over sizes and key ranges against torch.sort(stable=True).  python tools/sort_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for n in (10_000, 1_000_000, 30_000_000):
    for mx in (100, 60_000, 2_449_029, 2 ** 31 - 1, 2 ** 50):
        keys = torch.randint(0, mx, (n,), device=dev, generator=g)
        a = bench_legs._event_ms(lambda: ops.index_sort(keys, mx), 3, warmup=1)
        b = bench_legs._event_ms(lambda: ops.index_sort(keys), 3, warmup=1)
        t = bench_legs._event_ms(lambda: torch.sort(keys, stable=True), 3, warmup=1)
        print(f'n {n:9d} max {mx:16d}: index_sort(max_value) {a:7.3f} ms | index_sort() {b:7.3f} ms | torch.sort {t:7.3f} ms  ({t / a:.2f} x)', flush=True)
