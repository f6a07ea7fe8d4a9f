"""segment_sum_csr on the `skewed` degree distribution of tools/csr_shape_sweep.py (1000 hubs of 8000 positions = half of all
positions), a few calls: a driver for `rocprofv3 --kernel-trace --stats --output-format csv` (NOTES_r6 section 7)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyg_lib_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for dtype, K in ((torch.float32, 8), (torch.float32, 1)):
    E = 16_000_000
    N = E // 16
    deg = torch.poisson(torch.full((N,), 8.0, device=dev), generator=g).long()
    deg[::1000] += 8000
    ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), deg.cumsum(0)])
    E = int(ptr[-1])
    src = torch.randn(E, K, device=dev, generator=g).to(dtype)
    for _ in range(5):
        ops.segment_sum_csr(src, ptr)
    torch.cuda.synchronize()
