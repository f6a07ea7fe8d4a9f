"""C5 hetero sampler: single-batch loop against hetero_neighbor_sample_batched for several K (lanes: PYG_HIP_SAMPLER_LANES).
python tools/bench_hetero_batched.py [K ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import sampler  # noqa: E402

Ks = [int(a) for a in sys.argv[1:]] or [4, 8, 16]
dev = torch.device('cuda:0')
rp, cl = bench_legs.make_mag_graph(dev)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
fan = {e: [15, 10] for e in ets}
g = torch.Generator().manual_seed(1)
nb = 48
seeds = [torch.randperm(bench_legs.MAG_SIZES['paper'], generator=g)[:1024].to(dev) for _ in range(nb)]
torch.manual_seed(5)
for b in range(4):
    sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds[b]}, fan)
torch.cuda.synchronize()
t0 = time.perf_counter()
e = 0
for b in range(nb):
    e += sum(v.numel() for v in sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds[b]}, fan)[0].values())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
base = e / dt
print(f'lanes {os.environ.get("PYG_HIP_SAMPLER_LANES")}: single {dt / nb * 1e3:.3f} ms/batch, {base / 1e9:.2f} G edges/s', flush=True)
for K in Ks:
    calls = [[{'paper': seeds[(c * K + k) % nb]} for k in range(K)] for c in range(max(1, nb // K))]
    gs = [[1000 + c * K + k for k in range(K)] for c in range(len(calls))]
    sampler.hetero_neighbor_sample_batched(rp, cl, calls[0], fan, gs[0])
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e = 0
        for c in range(len(calls)):
            e += sum(sum(v.numel() for v in o[0].values()) for o in sampler.hetero_neighbor_sample_batched(rp, cl, calls[c], fan, gs[c]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f'  K={K}: {best / (len(calls) * K) * 1e3:.3f} ms/batch, {e / best / 1e9:.2f} G edges/s, x{e / best / base:.2f}', flush=True)
