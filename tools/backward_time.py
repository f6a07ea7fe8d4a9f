"""segment_matmul_backward + segment_k100_backward legs on one GPU: ms / frac only."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, bench_legs
dev = torch.device('cuda:0')
x, ptr, w, _ = bench.make_c2(dev, 0, 1)
r = bench_legs.leg_backward(dev, x, ptr, w)
print(json.dumps({k: r[k] for k in r if k in ('ms', 'frac', 'GBps')}))
del x, w
r = bench_legs.leg_grouped_mixed(dev)
print(json.dumps({k: {a: v[a] for a in v if a in ('ms', 'frac')} for k, v in r.items() if k in ('segment_k100', 'segment_k100_backward')}))
