import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler
from pyg_lib_amd import sampler
dev = torch.device('cuda:0')
rowptr, col = bench_sampler.make_graph(dev)
g = torch.Generator().manual_seed(1)
for i in range(12):
    seeds = torch.randperm(bench_sampler.N_NODES, generator=g)[:1024].to(dev)
    out = sampler.neighbor_sample(rowptr, col, seeds, [15, 10, 5])
torch.cuda.synchronize()
print(sampler.last_mode())
