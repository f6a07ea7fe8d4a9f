"""Host-side timeline of C3 sampler calls (PYG_HIP_SAMPLER_TRACE=1 prints it from the library) next to the wall time per call."""
import os, sys, time, torch
os.environ['PYG_HIP_SAMPLER_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler
from pyg_lib_amd import sampler
dev = torch.device('cuda:0')
rowptr, col = bench_sampler.make_graph(dev)
g = torch.Generator().manual_seed(1)
seeds = torch.randperm(bench_sampler.N_NODES, generator=g)[:1024 * 16].to(dev).view(16, 1024)
for b in range(16):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = sampler.neighbor_sample(rowptr, col, seeds[b], [15, 10, 5])
    torch.cuda.synchronize()
    print('call %d wall %.0f us' % (b, (time.perf_counter() - t) * 1e6), file=sys.stderr)
