"""General-shape forward kernel on K = 100 and on K = 128 (forced), a few launches each (for rocprofv3 --pmc passes):
    python tools/pmc_gen.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyg_lib_amd import ops
dev = torch.device('cuda', 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, B = 4_000_000, 47
ptr = torch.linspace(0, N, B + 1).long()
for K in (100, 128):
    x = torch.randn(N, K, device=dev).bfloat16()
    w = (torch.randn(B, K, 128, device=dev) / K ** 0.5).bfloat16()
    if K == 128:
        ops.set_matmul_schedule('general')
    for _ in range(iters):
        y = ops.segment_matmul(x, ptr, w)
    torch.cuda.synchronize()
    print(K, ops.matmul_last_variant())
    ops.set_matmul_schedule('auto')
    del x, w, y
