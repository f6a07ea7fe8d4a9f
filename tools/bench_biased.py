"""Biased (edge_weight) sampling on the C3 graph: python tools/bench_biased.py [batches] [cpu_batches]
Prints one JSON line: ms/batch, sampled edges/s, generator outputs drawn per batch, oracle (1 thread) rate."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler
from pyg_lib_amd import sampler

dev = torch.device('cuda:0')
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cpu_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rowptr, col = bench_sampler.make_graph(dev)
g = torch.Generator(device=dev).manual_seed(3)
w = torch.rand(col.numel(), device=dev, generator=g) + 0.05
gc = torch.Generator(device='cpu').manual_seed(1)
seeds = torch.randperm(bench_sampler.N_NODES, generator=gc)[:bench_sampler.BATCH * (batches + 3)].to(dev).view(-1, bench_sampler.BATCH)
for b in range(3):
    torch.manual_seed(12345)
    sampler.neighbor_sample(rowptr, col, seeds[b], bench_sampler.FANOUT, edge_weight=w)
torch.cuda.synchronize()
edges = 0
t0 = time.perf_counter()
for b in range(3, 3 + batches):
    torch.manual_seed(12345)
    out = sampler.neighbor_sample(rowptr, col, seeds[b], bench_sampler.FANOUT, edge_weight=w)
    edges += sum(out[5])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
res = dict(workload='biased neighbor_sample C3 graph, float32 weights', ms_per_batch=round(dt / batches * 1e3, 3),
           edges_per_s=round(edges / dt, 1), edges_per_batch=edges // batches)
if cpu_batches:
    import oracle
    rp, cl, wc = rowptr.cpu().numpy(), col.cpu().numpy(), w.cpu().numpy()
    t0 = time.perf_counter()
    ce = 0
    for b in range(cpu_batches):
        r = oracle.neighbor_sample(rp, cl, seeds[3 + b].cpu().numpy(), bench_sampler.FANOUT, edge_weight=wc, rng_seed=12345)
        ce += len(r[0])
        res['draws_per_batch'] = r[6]['rng_raw_draws']
    ct = time.perf_counter() - t0
    res['oracle_edges_per_s'] = round(ce / ct, 1)
    res['oracle_ms_per_batch'] = round(ct / cpu_batches * 1e3, 1)
    # parity on the bench graph itself
    torch.manual_seed(12345)
    o = sampler.neighbor_sample(rowptr, col, seeds[3 + cpu_batches - 1], bench_sampler.FANOUT, edge_weight=w)
    res['parity_last_cpu_batch'] = bool(np.array_equal(o[3].cpu().numpy(), r[3]) and np.array_equal(o[2].cpu().numpy(), r[2]))
print(json.dumps(res))
