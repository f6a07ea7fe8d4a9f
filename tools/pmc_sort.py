"""index_sort leg alone (for rocprofv3 passes):  python tools/pmc_sort.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
print(bench_legs.leg_index_sort(torch.device('cuda:0')))
