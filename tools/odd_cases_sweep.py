"""Odd corners for perf cliffs: many tiny segments, one / thousands of groups, many relations in the hetero sampler.
python tools/odd_cases_sweep.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops, sampler  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for rows, B in ((1_000_000, 100_000), (1_000_000, 10_000), (1_000_000, 1), (100, 1), (100, 50)):
    for dtype in (torch.bfloat16, torch.float32):
        cuts = torch.sort(torch.randint(0, rows, (B - 1,), device=dev, generator=g)).values if B > 1 else torch.empty(0, dtype=torch.long, device=dev)
        ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), cuts, torch.tensor([rows], device=dev)])
        x = torch.randn(rows, 128, device=dev, generator=g).to(dtype)
        w = (torch.randn(B, 128, 128, device=dev, generator=g) / 11).to(dtype)
        ms = bench_legs._event_ms(lambda: ops.segment_matmul(x, ptr, w), 5)
        alg = x.element_size() * (2 * rows * 128 + B * 128 * 128)
        print(f'segment_matmul rows {rows:8d} segments {B:6d} {str(dtype)[6:]:8s}: {ms:8.3f} ms  {alg / ms / 1e6:7.0f} GB/s  {ops.matmul_last_variant()}', flush=True)
        del x, w
for G in (1, 8, 5000):
    xs = [torch.randn(int(r), 128, device=dev, generator=g).bfloat16() for r in torch.randint(1, 2 * 600_000 // G, (G,)).tolist()]
    ws = [(torch.randn(128, 64, device=dev, generator=g) / 11).bfloat16() for _ in range(G)]
    ms = bench_legs._event_ms(lambda: ops.grouped_matmul(xs, ws), 3)
    n = sum(t.size(0) for t in xs)
    print(f'grouped_matmul {G:5d} groups, {n} rows, K=128 M=64: {ms:8.3f} ms  {2 * n * (128 + 64) / ms / 1e6:7.0f} GB/s  {ops.matmul_last_variant()}', flush=True)
    del xs, ws
# hetero sampler: many node types / relations
rng = np.random.default_rng(0)
for T, R in ((4, 7), (12, 40), (30, 120)):
    types = [f't{i}' for i in range(T)]
    sizes = {t: int(rng.integers(20_000, 200_000)) for t in types}
    ets = [(types[int(rng.integers(0, T))], f'r{i}', types[int(rng.integers(0, T))]) for i in range(R)]
    rp, cl = {}, {}
    for et in ets:
        d = rng.poisson(6, sizes[et[0]]).astype(np.int64)
        rp[et] = torch.from_numpy(np.concatenate([[0], np.cumsum(d)]).astype(np.int64)).to(dev)
        cl[et] = torch.from_numpy(rng.integers(0, sizes[et[2]], int(d.sum()), dtype=np.int64)).to(dev)
    fan = {e: [10, 5] for e in ets}
    seeds = [torch.from_numpy(rng.permutation(sizes[types[0]])[:1024].astype(np.int64)).to(dev) for _ in range(8)]
    torch.manual_seed(0)
    for s in seeds[:2]:
        sampler.hetero_neighbor_sample(rp, cl, {types[0]: s}, fan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = 0
    for s in seeds[2:]:
        e += sum(v.numel() for v in sampler.hetero_neighbor_sample(rp, cl, {types[0]: s}, fan)[0].values())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    print(f'hetero sampler {T:3d} types {R:4d} relations: {dt * 1e3:7.3f} ms per batch, {e / 6 / 1e3:7.1f} k edges, [{sampler.last_mode()}]', flush=True)
