"""auto vs general schedule over the (K, M) pairs around the dispatch boundary, long and short segments.
python tools/mm_shape_sweep2.py [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dtype = dict(bf16=torch.bfloat16, f16=torch.float16, f32=torch.float32)[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
esz = torch.empty(0, dtype=dtype).element_size()
shapes = [(256, 32), (256, 64), (256, 128), (256, 192), (256, 384), (256, 512), (512, 32), (512, 64), (512, 128), (512, 256), (512, 512),
          (128, 64), (64, 128), (128, 32), (64, 64), (128, 128)]
for kind in ('long', 'short', 'few'):
    for K, M in shapes:
        rows = (3_000_000 if K + M <= 512 else 1_500_000) if kind != 'few' else 40_000
        if kind == 'long':
            B = 47
            cuts = torch.sort(torch.randint(0, rows, (B - 1,), device=dev, generator=g)).values
            ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), cuts, torch.tensor([rows], device=dev)])
        elif kind == 'short':
            B = rows // 256
            ptr = torch.arange(0, B * 256 + 1, 256, device=dev)
            rows = B * 256
        else:
            B = 7
            ptr = torch.tensor([0, 5000, 5100, 20000, 20000, 33000, 39999, 40000], device=dev)
        x = torch.randn(rows, K, device=dev, generator=g).to(dtype)
        w = (torch.randn(B, K, M, device=dev, generator=g) / K ** 0.5).to(dtype)
        res = []
        for sched in ('auto', 'general'):
            ops.set_matmul_schedule(sched)
            ms = bench_legs._kernel_ms(lambda: ops.segment_matmul(x, ptr, w), iters=5, warmup=2)
            res.append((ms, ops.matmul_last_variant()))
        ops.set_matmul_schedule('auto')
        alg = esz * (rows * K + rows * M + B * K * M)
        print(f'{kind:5s} K={K:4d} M={M:4d}: auto {res[0][0]:7.3f} ms ({alg / res[0][0] / 8e6:.3f}, {res[0][1]})  general {res[1][0]:7.3f} ms ({alg / res[1][0] / 8e6:.3f})  '
              f'general/auto {res[1][0] / res[0][0]:.2f}', flush=True)
        del x, w
