"""Many sampler batches back to back (C3 homogeneous + C5 hetero): the one-launch scans' look-back must never hang and the
outputs of a repeated batch must stay identical.  python tools/stress_sampler.py [batches]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler, bench_legs
from pyg_lib_amd import sampler
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dev = torch.device('cuda:0')
rowptr, col = bench_sampler.make_graph(dev)
rp, cl = bench_legs.make_mag_graph(dev)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
fan = {e: [15, 10] for e in ets}
g = torch.Generator().manual_seed(7)
seeds = torch.randperm(bench_sampler.N_NODES, generator=g)[:1024].to(dev)
pseeds = torch.randperm(bench_legs.MAG_SIZES['paper'], generator=g)[:1024].to(dev)
torch.manual_seed(1)
ref = sampler.neighbor_sample(rowptr, col, seeds, [15, 10, 5])
torch.manual_seed(1)
href = sampler.hetero_neighbor_sample(rp, cl, {'paper': pseeds}, fan)
t = time.time()
bad = 0
for i in range(n):
    torch.manual_seed(1)
    out = sampler.neighbor_sample(rowptr, col, seeds, [15, 10, 5])
    if i % 50 == 0:
        bad += int(not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])))
    if i % 4 == 0:
        torch.manual_seed(1)
        h = sampler.hetero_neighbor_sample(rp, cl, {'paper': pseeds}, fan)
        if i % 200 == 0:
            bad += int(not all(torch.equal(h[2][k], href[2][k]) for k in href[2]))
torch.cuda.synchronize()
print('batches', n, 'hetero', (n + 3) // 4, 'mismatches', bad, 'seconds %.1f' % (time.time() - t))
