import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench_legs
from pyg_lib_amd import sampler
dev = torch.device('cuda:0')
rp, cl = bench_legs.make_mag_graph(dev)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
fan = {e: [15, 10] for e in ets}
g = torch.Generator().manual_seed(1)
seeds = [torch.randperm(bench_legs.MAG_SIZES['paper'], generator=g)[:1024].to(dev) for _ in range(24)]
for s in seeds[:8]:
    sampler.hetero_neighbor_sample(rp, cl, {'paper': s}, fan)
torch.cuda.synchronize()
for s in seeds[8:]:
    t = time.perf_counter()
    out = sampler.hetero_neighbor_sample(rp, cl, {'paper': s}, fan)
    dt = (time.perf_counter() - t) * 1e6
    print('python front total %.1f us' % dt, file=sys.stderr, flush=True)
    del out
