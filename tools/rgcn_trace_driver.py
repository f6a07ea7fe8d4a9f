"""A few C5 hetero sampler batches for kernel tracing (tools/sampler_trace.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.bench_rgcn as br
from pyg_lib_amd import sampler
dev = torch.device('cuda', 0)
rp, cl = br.make_graph(dev)
ets = [(s, r, d) for s, r, d, _ in br.RELS]
fan = {e: [15, 10] for e in ets}
gs = torch.Generator().manual_seed(1)
for i in range(6):
    seeds = torch.randperm(br.SIZES['paper'], generator=gs)[:1024].to(dev)
    torch.manual_seed(100 + i)
    sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds}, fan)
torch.cuda.synchronize()
