"""segment_sum_csr / segment_max_csr over narrow rows and row lengths: which kernel should take them?  ms for 16 M positions.
python tools/narrow_row_kernels.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
from pyg_lib_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for dtype, K in ((torch.float32, 1), (torch.float32, 2), (torch.float32, 3), (torch.float32, 5), (torch.bfloat16, 1), (torch.bfloat16, 2), (torch.bfloat16, 4), (torch.float32, 4), (torch.float32, 8), (torch.float32, 12), (torch.bfloat16, 8), (torch.bfloat16, 16), (torch.bfloat16, 24)):
    for mean_deg in (2, 4, 8, 16, 48):
        E = 16_000_000
        N = E // mean_deg
        deg = torch.poisson(torch.full((N,), float(mean_deg), device=dev), generator=g).long()
        ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), deg.cumsum(0)])
        E = int(ptr[-1])
        src = torch.randn(E, K, device=dev, generator=g).to(dtype)
        a = bench_legs._event_ms(lambda: ops.segment_sum_csr(src, ptr), 5, warmup=2)
        b = bench_legs._event_ms(lambda: ops.segment_max_csr(src, ptr), 5, warmup=2)
        small = src[:N].contiguous()
        c = bench_legs._event_ms(lambda: ops.gather_csr(small, ptr), 5, warmup=2)
        print(f'{str(dtype)[6:]:9s} K={K:3d} deg {mean_deg:3d}: sum {a:.3f} | max {b:.3f} | gather {c:.3f} ms', flush=True)
