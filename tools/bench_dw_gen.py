"""Weight-gradient kernels side by side: the shape-specialised kernel, and the general-shape kernel (matmul_dw_gen.hip)
on the same shape (an 8-byte storage offset sends K = M = 128 through it), on K = 100 (8-byte rows, tail chunk), K = 96
(16-byte rows, no tail) and fp32.  Prints ms and TB/s of X + dY read once."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyg_lib_amd import ops  # noqa: E402

DEV = 'cuda:0'


def timed(fn, iters=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def case(name, n, K, M, dtype, B=47, offset_elems=0):
    g = torch.Generator(device=DEV).manual_seed(1)
    bx = torch.randn(n * K + 64, device=DEV, generator=g).to(dtype)
    by = torch.randn(n * M + 64, device=DEV, generator=g).to(dtype)
    x = bx[offset_elems:offset_elems + n * K].view(n, K)
    gy = by[offset_elems:offset_elems + n * M].view(n, M)
    fr = torch.rand(B)
    sizes = torch.floor(fr / fr.sum() * n).long()
    sizes[-1] += n - sizes.sum()
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    before = ops.matmul_dw_counters()
    ms = timed(lambda: torch.ops.pyg.segment_matmul_grad_other(x, ptr, gy))
    after = ops.matmul_dw_counters()
    kern = 'gen' if after[1] > before[1] else 'fast'
    s = x.element_size()
    byt = s * n * (K + M)
    print(f'{name:42s} {kern:5s} {ms:8.3f} ms  {byt / ms / 1e9:6.2f} TB/s', flush=True)


if __name__ == '__main__':
    n = 8_000_000
    case('bf16 K=128 M=128 aligned', n, 128, 128, torch.bfloat16)
    case('bf16 K=128 M=128 8-byte offset', n, 128, 128, torch.bfloat16, offset_elems=4)
    case('bf16 K=128 M=128 2-byte offset', n, 128, 128, torch.bfloat16, offset_elems=1)
    case('bf16 K=100 M=128', n, 100, 128, torch.bfloat16)
    case('bf16 K=96 M=128', n, 96, 128, torch.bfloat16)
    case('bf16 K=100 M=47', n, 100, 47, torch.bfloat16)
    case('bf16 K=256 M=256 8-byte offset', n // 2, 256, 256, torch.bfloat16, offset_elems=4)
    case('f32 K=128 M=128 aligned', n // 2, 128, 128, torch.float32)
    case('f32 K=100 M=128', n // 2, 100, 128, torch.float32)
    case('f32 K=128 M=128 4-byte offset', n // 2, 128, 128, torch.float32, offset_elems=1)
