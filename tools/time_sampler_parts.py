import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler
from pyg_lib_amd import sampler
dev = torch.device('cuda:0')
rowptr, col = bench_sampler.make_graph(dev)
g = torch.Generator(device='cpu').manual_seed(1)
seeds = torch.randperm(bench_sampler.N_NODES, generator=g)[:1024 * 30].to(dev).view(30, 1024)
for b in range(3):
    sampler.neighbor_sample(rowptr, col, seeds[b], [15, 10, 5])
torch.cuda.synchronize()
def T(f, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): f(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print('manual_seed ms', T(lambda i: torch.manual_seed(12345)))
print('op only ms', T(lambda i: sampler.neighbor_sample(rowptr, col, seeds[3 + i], [15, 10, 5])))
def both(i):
    torch.manual_seed(12345)
    return sampler.neighbor_sample(rowptr, col, seeds[3 + i], [15, 10, 5])
print('seed+op ms', T(both))
print('raw torch.ops ms', T(lambda i: torch.ops.pyg.neighbor_sample(rowptr, col, seeds[3 + i], [15, 10, 5])))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): both(i)
pr.disable(); pstats.Stats(pr).sort_stats('cumtime').print_stats(8)
