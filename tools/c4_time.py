"""C4 (grouped_matmul, 512 groups, K = M = 256, bf16) through bench_legs.leg_c4 on one GPU: kernel ms / frac only."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
r = bench_legs.leg_c4(torch.device('cuda:0'), 0, 1)
print(json.dumps({k: r[k] for k in r if k in ('compute_only', 'rank0_launch', 'operator_ms', 'variant')}))
