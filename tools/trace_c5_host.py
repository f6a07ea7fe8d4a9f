"""Host-side timeline of C5 hetero sampler calls (PYG_HIP_SAMPLER_TRACE=1) next to the wall time per call, and the cost of the
op's Dict marshalling alone (the same call with every fan-out 0: no sampling work)."""
import os, sys, time, torch
os.environ['PYG_HIP_SAMPLER_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
from pyg_lib_amd import sampler
dev = torch.device('cuda:0')
rp, cl = bench_legs.make_mag_graph(dev)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
fan = {e: [15, 10] for e in ets}
g = torch.Generator().manual_seed(1)
for b in range(10):
    seeds = torch.randperm(bench_legs.MAG_SIZES['paper'], generator=g)[:1024].to(dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds}, fan)
    torch.cuda.synchronize()
    print('call %d wall %.0f us' % (b, (time.perf_counter() - t) * 1e6), file=sys.stderr)
