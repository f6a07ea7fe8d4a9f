"""neighbor_sample on the C3 graph over batch sizes, fan-outs and modes: ms per batch, sampled edges/s, the driver that ran.
python tools/sampler_sweep.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler  # noqa: E402
from pyg_lib_amd import sampler  # noqa: E402

dev = torch.device('cuda:0')
rowptr, col = bench_sampler.make_graph(dev)
g = torch.Generator().manual_seed(1)
w = torch.rand(col.numel(), device=dev)
cases = []
for batch in (64, 1024, 8192):
    for fan in ([15, 10, 5], [25, 10], [10, 10, 10], [5], [64, 32]):
        cases.append((batch, fan, {}))
cases += [(1024, [15, 10, 5], dict(replace=True)), (1024, [15, 10, 5], dict(disjoint=True)), (1024, [15, 10, 5], dict(return_edge_id=False)),
          (256, [-1, -1], {}), (1024, [15, 10, 5], dict(edge_weight=w)), (1024, [15, 10, 5], dict(csc=True))]
for batch, fan, kw in cases:
    seeds = [torch.randperm(bench_sampler.N_NODES, generator=g)[:batch].to(dev) for _ in range(12)]
    torch.manual_seed(3)
    for s in seeds[:3]:
        sampler.neighbor_sample(rowptr, col, s, fan, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = 0
    for s in seeds[3:]:
        e += sum(sampler.neighbor_sample(rowptr, col, s, fan, **kw)[5])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 9
    print(f'batch {batch:5d} fan-out {str(fan):14s} {str({k: (v if not torch.is_tensor(v) else "tensor") for k, v in kw.items()}):28s}: {dt * 1e3:7.3f} ms, '
          f'{e / 9 / 1e3:8.1f} k edges per batch, {e / 9 / dt / 1e9:5.2f} G edges/s  [{sampler.last_mode()}]', flush=True)
