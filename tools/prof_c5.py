"""C5 hetero sampling batches for rocprofv3 --kernel-trace (tools/trace_batch.py prints the last one)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
from pyg_lib_amd import sampler
dev = torch.device('cuda:0')
rp, cl = bench_legs.make_mag_graph(dev)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
fan = {e: [15, 10] for e in ets}
g = torch.Generator().manual_seed(1)
for i in range(12):
    seeds = torch.randperm(bench_legs.MAG_SIZES['paper'], generator=g)[:1024].to(dev)
    out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds}, fan)
torch.cuda.synchronize()
print(sampler.last_mode())
