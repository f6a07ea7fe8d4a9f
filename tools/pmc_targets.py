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
del src, feat, index, sidx, indptr
# round 3: general-shape grouped matmul (mixed K), hetero sample (fused chain) + R-GCN layer from the global tables
import math
import bench_legs
from pyg_lib_amd import rgcn
g2 = torch.Generator().manual_seed(0)
rows = torch.exp(torch.rand(64, generator=g2) * (math.log(262144.0) - math.log(1024.0)) + math.log(1024.0))
rows = (rows / rows.sum() * 6_000_000).long().tolist()
kk = [(100, 128, 256, 768)[i % 4] for i in range(64)]
xs = [torch.randn(r, k, device=dev, generator=gd).bfloat16() for r, k in zip(rows, kk)]
ws = [(torch.randn(k, 128, device=dev, generator=gd) / k ** 0.5).bfloat16() for k in kk]
for _ in range(3):
    ops.grouped_matmul(xs, ws)
del xs, ws
# round 3: C4 (K = M = 256, register-W kernel) and fp32 K = 128 (split-bf16 kernel, 4 Mi rows in 64 segments)
c4rows = bench_legs.c4_group_rows()
xs = [torch.randn(r, 256, device=dev, generator=gd).bfloat16() for r in c4rows]
ws = [(torch.randn(256, 256, device=dev, generator=gd) / 16).bfloat16() for _ in c4rows]
for _ in range(3):
    ops.grouped_matmul(xs, ws)
del xs, ws
xf = torch.randn(1 << 22, 128, device=dev, generator=gd)
wf = torch.randn(64, 128, 128, device=dev, generator=gd) / 11
pf = torch.arange(0, (1 << 22) + 1, (1 << 22) // 64)
for _ in range(3):
    ops.segment_matmul(xf, pf, wf)
del xf, wf
types = list(bench_legs.MAG_SIZES)
ets = [(s_, r_, d_) for s_, r_, d_, _ in bench_legs.MAG_RELS]
rp, cl = bench_legs.make_mag_graph(dev)
featm = {t: torch.randn(bench_legs.MAG_SIZES[t], 128, device=dev, generator=gd).bfloat16() for t in types}
Wm = (torch.randn(len(ets), 128, 128, device=dev, generator=gd) / 11).bfloat16()
fan = {e: [15, 10] for e in ets}
for b in range(4):
    sd = torch.randperm(bench_legs.MAG_SIZES['paper'], generator=g)[:1024].to(dev)
    out = sampler.hetero_neighbor_sample(rp, cl, {'paper': sd}, fan)
    rgcn.rgcn_layer_fused_tables(featm, out[2], types, out[0], out[1], ets, Wm, grouped=False)  # atomic kernel (+ zero fill)
    rgcn.rgcn_layer_fused_tables(featm, out[2], types, out[0], out[1], ets, Wm, grouped=True)   # row-start + owner-computes kernel
torch.cuda.synchronize()
del rp, cl, featm
keys = torch.randint(0, 2_449_029, (100_000_000,), device=dev, generator=gd)
for _ in range(3):
    ops.index_sort(keys, 2_449_029)
torch.cuda.synchronize()
