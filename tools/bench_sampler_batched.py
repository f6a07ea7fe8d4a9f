"""C3 sampler: single-batch loop against neighbor_sample_batched for several K (lanes: PYG_HIP_SAMPLER_LANES, HW queues:
GPU_MAX_HW_QUEUES).   python tools/bench_sampler_batched.py [K ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler  # noqa: E402
from pyg_lib_amd import sampler  # noqa: E402

if __name__ == '__main__':
    Ks = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]
    dev = torch.device('cuda:0')
    rowptr, col = bench_sampler.make_graph(dev)
    g = torch.Generator(device='cpu').manual_seed(1)
    nb = 64
    seeds = torch.randperm(bench_sampler.N_NODES, generator=g)[:bench_sampler.BATCH * nb].to(dev).view(nb, -1)
    torch.manual_seed(12345)
    for b in range(3):
        sampler.neighbor_sample(rowptr, col, seeds[b], bench_sampler.FANOUT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = 0
    for b in range(nb):
        e += sum(sampler.neighbor_sample(rowptr, col, seeds[b], bench_sampler.FANOUT)[5])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    base = e / dt
    print(f'lanes {os.environ.get("PYG_HIP_SAMPLER_LANES")} queues {os.environ.get("GPU_MAX_HW_QUEUES")}: single {dt / nb * 1e3:.3f} ms/batch, '
          f'{base / 1e9:.2f} G edges/s', flush=True)
    for K in Ks:
        lists = [[seeds[(c * K + k) % nb] for k in range(K)] for c in range(max(1, nb // K))]
        gs = [[1000 + c * K + k for k in range(K)] for c in range(len(lists))]
        sampler.neighbor_sample_batched(rowptr, col, lists[0], bench_sampler.FANOUT, gs[0])
        best = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e = 0
            for c in range(len(lists)):
                e += sum(sum(o[5]) for o in sampler.neighbor_sample_batched(rowptr, col, lists[c], bench_sampler.FANOUT, gs[c]))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(f'  K={K}: {best / (len(lists) * K) * 1e3:.3f} ms/batch, {e / best / 1e9:.2f} G edges/s, x{e / best / base:.2f}', flush=True)
