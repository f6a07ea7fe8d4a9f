"""segment_matmul over common (K, M) pairs: ms, algorithmic GB/s, fraction of 8 TB/s, kernel variant.
python tools/mm_shape_sweep.py [rows] [dtype] [schedule: auto | general]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
dtype = dict(bf16=torch.bfloat16, f16=torch.float16, f32=torch.float32)[sys.argv[2] if len(sys.argv) > 2 else 'bf16']
dev = torch.device('cuda:0')
ops.set_matmul_schedule(sys.argv[3] if len(sys.argv) > 3 else 'auto')
g = torch.Generator(device=dev).manual_seed(0)
B = 47
cuts = torch.sort(torch.randint(0, rows, (B - 1,), device=dev, generator=g)).values
ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), cuts, torch.tensor([rows], device=dev)])
esz = torch.empty(0, dtype=dtype).element_size()
for K, M in [(128, 128), (64, 64), (32, 32), (16, 16), (128, 64), (64, 128), (128, 256), (256, 128), (256, 256), (256, 64), (64, 256),
             (512, 128), (128, 512), (512, 512), (768, 128), (100, 128), (128, 100), (96, 96), (192, 192), (300, 300), (128, 32), (128, 8)]:
    n = rows if K + M <= 512 else rows // 2
    x = torch.randn(n, K, device=dev, generator=g).to(dtype)
    w = (torch.randn(B, K, M, device=dev, generator=g) / K ** 0.5).to(dtype)
    p = ptr if n == rows else (ptr // 2)
    f = lambda: ops.segment_matmul(x, p, w)
    ms = bench_legs._kernel_ms(f, iters=5, warmup=2)
    alg = esz * (n * K + n * M + B * K * M)
    print(f'K={K:4d} M={M:4d}: {ms:7.3f} ms  {alg / ms / 1e6:7.0f} GB/s  frac {alg / ms / 1e6 / 8000:.3f}  {2.0 * n * K * M / ms / 1e9:6.0f} TFLOP/s  {ops.matmul_last_variant()}', flush=True)
    del x, w
