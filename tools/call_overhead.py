import torch, time, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from pyg_lib_amd import sampler
dev='cuda:0'
types=['a','b','c','d']
ets=[('a','r0','b'),('b','r1','a'),('a','r2','a'),('a','r3','c'),('c','r4','a'),('a','r5','d'),('d','r6','a')]
n=1000
rp={e: torch.arange(0, n+1, device=dev)*0 for e in ets}
cl={e: torch.zeros(0, dtype=torch.long, device=dev) for e in ets}
seeds={'a': torch.arange(10, device=dev)}
fan={e:[15,10] for e in ets}
for _ in range(5): sampler.hetero_neighbor_sample(rp, cl, seeds, fan)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(200): sampler.hetero_neighbor_sample(rp, cl, seeds, fan)
torch.cuda.synchronize(); print('hetero empty-graph call us', (time.perf_counter()-t)/200*1e6)
rp1=torch.zeros(n+1, dtype=torch.long, device=dev); cl1=torch.zeros(0, dtype=torch.long, device=dev)
for _ in range(5): sampler.neighbor_sample(rp1, cl1, seeds['a'], [15,10])
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(200): sampler.neighbor_sample(rp1, cl1, seeds['a'], [15,10])
torch.cuda.synchronize(); print('homo empty-graph call us', (time.perf_counter()-t)/200*1e6)
