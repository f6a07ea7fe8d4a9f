"""C4 grouped_matmul leg alone (for rocprofv3 --pmc passes):  python tools/pmc_c4.py [iters]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
print(bench_legs.leg_c4(torch.device('cuda:0'), 0, 1, iters=it))
