import torch, time, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import bench_sampler
from pyg_lib_amd import sampler
dev = torch.device('cuda:0')
rowptr, col = bench_sampler.make_graph(dev)
w = torch.rand(col.numel(), device=dev) + 0.05
seeds = torch.randperm(bench_sampler.N_NODES)[:1024].to(dev)
torch.manual_seed(1); sampler.neighbor_sample(rowptr, col, seeds, [15, 10, 5]); torch.cuda.synchronize()
for i in range(3):
    torch.manual_seed(1); t = time.perf_counter()
    sampler.neighbor_sample(rowptr, col, seeds, [15, 10, 5], edge_weight=w); torch.cuda.synchronize()
    print('biased call', i, round((time.perf_counter() - t) * 1e3, 2), 'ms')
