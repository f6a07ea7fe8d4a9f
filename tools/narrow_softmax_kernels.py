"""softmax_csr (forward + backward) over narrow inner sizes and group lengths: which kernel should take them?  ms for 16 M positions.
python tools/narrow_softmax_kernels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for K in (1, 2, 4, 8, 16):
    for mean_deg in (2, 8, 16, 48):
        E = 16_000_000 // max(1, K // 4)
        N = E // mean_deg
        deg = torch.poisson(torch.full((N,), float(mean_deg), device=dev), generator=g).long()
        ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), deg.cumsum(0)])
        E = int(ptr[-1])
        src = torch.randn(E, K, device=dev, generator=g)
        a = bench_legs._event_ms(lambda: ops.softmax_csr(src, ptr), 5, warmup=2)
        x = src.clone().requires_grad_()
        y = ops.softmax_csr(x, ptr)
        go = torch.randn_like(y)
        b = bench_legs._event_ms(lambda: torch.autograd.grad(y, x, go, retain_graph=True), 5, warmup=2)
        print(f'float32 inner={K:3d} {mean_deg:3d} per group, E={E}: forward {a:.3f} | backward {b:.3f} ms', flush=True)
