"""forward + backward of segment_matmul over (K, M) pairs: forward ms, backward ms (dX + dW), the kernels' names.
python tools/mm_backward_sweep.py [rows] [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
dtype = dict(bf16=torch.bfloat16, f16=torch.float16, f32=torch.float32)[sys.argv[2] if len(sys.argv) > 2 else 'bf16']
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
B = 47
cuts = torch.sort(torch.randint(0, rows, (B - 1,), device=dev, generator=g)).values
ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), cuts, torch.tensor([rows], device=dev)])
esz = torch.empty(0, dtype=dtype).element_size()
for K, M in [(128, 128), (64, 64), (128, 64), (64, 128), (256, 256), (256, 128), (128, 256), (256, 64), (64, 256), (512, 128), (128, 512),
             (512, 512), (100, 128), (192, 192), (32, 32)]:
    n = rows if K + M <= 512 else rows // 2
    p = ptr if n == rows else ptr // 2
    x = torch.randn(n, K, device=dev, generator=g).to(dtype).requires_grad_()
    w = (torch.randn(B, K, M, device=dev, generator=g) / K ** 0.5).to(dtype).requires_grad_()
    go = torch.randn(n, M, device=dev, generator=g).to(dtype)
    fwd = bench_legs._event_ms(lambda: ops.segment_matmul(x, p, w), 4)

    def fb():
        x.grad = w.grad = None
        ops.segment_matmul(x, p, w).backward(go)

    both = bench_legs._event_ms(fb, 4)
    alg = esz * n * (K + M)
    print(f'K={K:4d} M={M:4d}: forward {fwd:7.3f} ms ({alg / fwd / 8e6:.3f} of HBM)  backward {both - fwd:7.3f} ms ({2 * alg / max(both - fwd, 1e-9) / 8e6:.3f})', flush=True)
    del x, w, go
