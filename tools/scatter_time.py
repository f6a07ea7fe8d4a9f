"""scatter_sum / segment_sum_coo / gather_coo legs of the bench on one GPU (bench_legs.leg_scatter_sum)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
r = bench_legs.leg_scatter_sum(torch.device('cuda:0'))
print(json.dumps({k: (v if not isinstance(v, dict) else {a: v[a] for a in v if a in ('ms', 'frac', 'GBps')}) for k, v in r.items()
                  if k in ('ms', 'frac', 'GBps', 'segment_sum_coo_sorted', 'gather_coo')}))
