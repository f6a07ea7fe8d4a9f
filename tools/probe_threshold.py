"""bf16 F = 128, 21 M rows cut into equal segments: ticket kernel vs item ring (where is the crossover?)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyg_lib_amd import ops
from bench_legs import _kernel_ms
dev = 'cuda:0'
n = 21_000_000 // 65536 * 65536
x = torch.randn(n, 128, device=dev).bfloat16()
for rows in (2048, 4096, 8192, 16384, 32768, 65536):
    B = n // rows
    w = (torch.randn(B, 128, 128, device=dev) / 11).bfloat16()
    ptr = torch.arange(0, n + 1, rows)
    r = {}
    for sched in ('ticket', 'ring'):
        ops.set_matmul_schedule(sched)
        r[sched] = _kernel_ms(lambda: ops.segment_matmul(x, ptr, w), iters=5, warmup=2)
    print(rows, 'rows per segment (B = %d): ticket %.3f ms, ring %.3f ms' % (B, r['ticket'], r['ring']))
    del w
ops.set_matmul_schedule('auto')
