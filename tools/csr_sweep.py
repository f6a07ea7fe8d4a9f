import sys, os, torch
sys.path.insert(0, '/root/repo')
import bench_legs
r = bench_legs.leg_scatter_sum(torch.device('cuda:0'))
print('sorted', r['segment_sum_coo_sorted'], 'unsorted', r['ms'], r['frac'], 'gather', r['gather_coo']['frac'], flush=True)
