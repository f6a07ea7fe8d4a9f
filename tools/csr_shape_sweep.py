"""CSR family over row widths and degree distributions: ms and fraction of 8 TB/s on the algorithmic bytes (src read + out written),
next to torch.segment_reduce where it exists.   python tools/csr_shape_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    esz = torch.empty(0, dtype=dtype).element_size()
    for K in (1, 8, 64, 128):
        for deg_name, mean_deg in (('deg 2', 2), ('deg 16', 16), ('deg 300', 300), ('skewed', -1)):
            E = max(500_000, min(16_000_000, 1_500_000_000 // (K * esz)))
            if mean_deg > 0:
                N = max(1, E // mean_deg)
                deg = torch.poisson(torch.full((N,), float(mean_deg), device=dev), generator=g).long()
            else:
                N = E // 16
                deg = torch.poisson(torch.full((N,), 8.0, device=dev), generator=g).long()
                deg[::1000] += 8000   # a hub every 1000 rows
            ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), deg.cumsum(0)])
            E = int(ptr[-1])
            src = torch.randn(E, K, device=dev, generator=g).to(dtype)
            alg = esz * (E * K + N * K) + 8 * N
            res = []
            for name, fn in (('sum', lambda: ops.segment_sum_csr(src, ptr)), ('max', lambda: ops.segment_max_csr(src, ptr)),
                             ('softmax', lambda: ops.softmax_csr(src, ptr) if dtype == torch.float32 else None),
                             ('gather', lambda: ops.gather_csr(src[:N], ptr)),
                             ('torch.segment_reduce', lambda: torch.segment_reduce(src, 'sum', lengths=deg, axis=0, unsafe=True) if dtype == torch.float32 else None)):
                if fn() is None:
                    continue
                ms = bench_legs._event_ms(fn, 3, warmup=1)
                a = alg * (2 if name == 'softmax' else 1)
                res.append(f'{name} {ms:.3f} ms ({a / ms / 8e9:.2f})')
            print(f'{str(dtype)[6:]:9s} K={K:3d} {deg_name:8s} E={E:9d}: ' + ' | '.join(res), flush=True)
            del src
