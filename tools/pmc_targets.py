"""Workloads for PMC passes (FETCH_SIZE / WRITE_SIZE per kernel): a few C3 sampler batches, gather_coo,
index_sort, segment_sum_csr, scatter_max.  Run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d out_f -- python tools/pmc_targets.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d out_w -- python tools/pmc_targets.py
and summarise with tools/pmc_summary.py out_f out_w."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler
from pyg_lib_amd import ops, sampler
dev = torch.device('cuda:0')
rowptr, col = bench_sampler.make_graph(dev)
g = torch.Generator(device='cpu').manual_seed(1)
seeds = torch.randperm(bench_sampler.N_NODES, generator=g)[:1024 * 6].to(dev).view(6, 1024)
for b in range(6):
    torch.manual_seed(12345)
    sampler.neighbor_sample(rowptr, col, seeds[b], [15, 10, 5])
torch.cuda.synchronize()
gd = torch.Generator(device=dev).manual_seed(0)
E, N, K = 20_000_000, 2_000_000, 128
feat = torch.randn(N, K, device=dev, generator=gd).bfloat16()
index = torch.randint(0, N, (E,), device=dev, generator=gd)
sidx = torch.sort(index).values
for _ in range(3):
    ops.gather_coo(feat, sidx)
    ops.gather_coo(feat, index)
src = torch.randn(E, K, device=dev, generator=gd).bfloat16()
indptr = torch.zeros(N + 1, dtype=torch.long, device=dev)
indptr[1:] = torch.bincount(sidx, minlength=N).cumsum(0)
for _ in range(3):
    ops.segment_sum_csr(src, indptr)
    ops.scatter_max(src, index, 0, None, N)
keys = torch.randint(0, 2_449_029, (100_000_000,), device=dev, generator=gd)
for _ in range(3):
    ops.index_sort(keys, 2_449_029)
torch.cuda.synchronize()
