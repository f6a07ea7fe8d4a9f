"""The fused R-GCN layer on a graph that was not sampled (no fan-out bound): one destination node of one relation collects a share
of the edges.  ms of rgcn_layer_fused (grouped / atomic) and of the three-op chain: without a hub | 2.5 % | 25 % of the edges.
python tools/hub_sweep_layer.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import rgcn  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
N, E, R, F = 200_000, 2_000_000, 4, 128
types = ['a']
off = rgcn.type_offsets({'a': N}, types)
x = torch.randn(N, F, device=dev, generator=g).bfloat16()
W = (torch.randn(R, F, F, device=dev, generator=g) / F ** 0.5).bfloat16()
ets = [('a', f'r{i}', 'a') for i in range(R)]
res = {}
for hub in (0, 50_000, 500_000):
    rows, cols = {}, {}
    for i, et in enumerate(ets):
        r = torch.randint(0, N, (E // R,), device=dev, generator=g)
        if hub and i == 1:
            r[torch.randperm(E // R, device=dev, generator=g)[:hub]] = 4321
        rows[et] = torch.sort(r).values
        cols[et] = torch.randint(0, N, (E // R,), device=dev, generator=g)
    for name, fn in (('fused, atomic-free', lambda: rgcn.rgcn_layer_fused(x, off, rows, cols, ets, W, grouped=True)),
                     ('fused, atomics', lambda: rgcn.rgcn_layer_fused(x, off, rows, cols, ets, W, grouped=False)),
                     ('three-op chain', lambda: rgcn.rgcn_layer(x, off, rows, cols, ets, W))):
        res.setdefault(name, []).append(bench_legs._event_ms(fn, 3, warmup=1))
        res.setdefault(name + ' path', []).append(rgcn.last_layer_path())
for name, v in res.items():
    if name.endswith('path'):
        continue
    print(f'{name:20s} {v[0]:9.3f} | {v[1]:9.3f} | {v[2]:9.3f} ms   ({res[name + " path"][0]})', flush=True)
a = rgcn.rgcn_layer_fused(x, off, rows, cols, ets, W, grouped=True).float()
b = rgcn.rgcn_layer(x, off, rows, cols, ets, W).float()
print('hub row, atomic-free vs chain: max abs diff', float((a[4321] - b[4321]).abs().max()), 'of', float(b[4321].abs().max()))
