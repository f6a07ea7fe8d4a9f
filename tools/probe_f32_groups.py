"""fp32 K = M = 128 segment_matmul, 4 Mi rows cut into B equal segments: the two split-bf16 kernels side by side."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyg_lib_amd import ops
from bench_legs import _kernel_ms
dev = 'cuda:0'
n = 1 << 22
x = torch.randn(n, 128, device=dev)
for B in (16, 256, 4096, 32768):
    w = torch.randn(B, 128, 128, device=dev) / 11
    ptr = torch.arange(0, n + 1, n // B)
    for sched in ('auto', 'contiguous'):
        ops.set_matmul_schedule(sched)
        ms = _kernel_ms(lambda: ops.segment_matmul(x, ptr, w), iters=8, warmup=2)
        print(B, 'segments of', n // B, 'rows:', ops.matmul_last_variant(), '%.3f ms' % ms, '%.2f TB/s' % ((n * 1024 + B * 65536) / ms / 1e9))
ops.set_matmul_schedule('auto')
