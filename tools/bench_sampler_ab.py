import sys, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler, bench_legs
from pyg_lib_amd import sampler
dev = torch.device('cuda:0')
r = bench_sampler.run(dev, cpu_batches=0) if 'cpu_batches' in bench_sampler.run.__code__.co_varnames else bench_sampler.run(dev)
print('mode', sampler.last_mode(), json.dumps({k: r[k] for k in ('ms_per_batch', 'value', 'edges_per_batch')}))
c5 = bench_legs.leg_c5(dev)
print('mode', sampler.last_mode(), json.dumps({k: c5[k] for k in ('ms_end_to_end', 'ms_sampler', 'edges_per_batch', 'layer')}))
