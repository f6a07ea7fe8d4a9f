"""index_sort over key DISTRIBUTIONS (the radix passes count digits in LDS: equal keys contend for one counter): ms for 50 M keys,
against torch.sort(stable=True); results checked.   python tools/sort_distributions.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
n, top = 50_000_000, 2_449_029
dists = {
    'uniform': torch.randint(0, top, (n,), device=dev, generator=g),
    'all equal': torch.full((n,), 12345, device=dev),
    'two values': torch.randint(0, 2, (n,), device=dev, generator=g) * (top - 1),
    'sorted': torch.arange(n, device=dev) // 21,
    'reversed': (n - 1 - torch.arange(n, device=dev)) // 21,
    '25 % one key': torch.where(torch.rand(n, device=dev, generator=g) < 0.25, torch.tensor(777, device=dev),
                                torch.randint(0, top, (n,), device=dev, generator=g)),
    'zipf-like': (torch.rand(n, device=dev, generator=g).pow(8) * top).long(),
}
for name, keys in dists.items():
    ms = bench_legs._event_ms(lambda: ops.index_sort(keys, top), 3, warmup=1)
    mt = bench_legs._event_ms(lambda: torch.sort(keys, stable=True), 2, warmup=1)
    out, perm = ops.index_sort(keys, top)
    ref, rperm = torch.sort(keys, stable=True)
    ok = torch.equal(out, ref) and torch.equal(perm, rperm)
    print(f'{name:14s}: index_sort {ms:7.3f} ms | torch.sort(stable) {mt:7.3f} ms | {"exact" if ok else "MISMATCH"}', flush=True)
