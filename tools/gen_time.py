"""grouped_mixed / segment_k100 legs of the bench on one GPU (bench_legs.leg_grouped_mixed)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
r = bench_legs.leg_grouped_mixed(torch.device('cuda:0'))
print(json.dumps({k: (v if not isinstance(v, dict) else {a: v[a] for a in v if a in ('ms', 'frac', 'GBps')}) for k, v in r.items()
                  if k in ('ms', 'frac', 'GBps', 'segment_k100', 'segment_k100_backward')}))
