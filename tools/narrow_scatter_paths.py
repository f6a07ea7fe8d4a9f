"""Narrow unsorted scatter_sum rows: the atomic kernels against the stable-sort + CSR-row path (what torch.use_deterministic_algorithms
selects) over sizes, uniform and with a 0.25 % hub.   python tools/narrow_scatter_paths.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
for dtype, K in ((torch.float32, 1), (torch.float32, 4), (torch.bfloat16, 16), (torch.float32, 12)):
    for E in (65_536, 262_144, 1_048_576, 4_194_304, 16_777_216):
        N = E // 8
        src = torch.randn(E, K, device=dev, generator=g).to(dtype)
        out = []
        for hub in (0, E // 400):
            idx = torch.randint(0, N, (E,), device=dev, generator=g)
            if hub:
                idx[torch.randperm(E, device=dev, generator=g)[:hub]] = 7
            for det in (False, True):
                torch.use_deterministic_algorithms(det)
                out.append(bench_legs._event_ms(lambda: ops.scatter_sum(src, idx, 0, None, N), 5, warmup=2))
            torch.use_deterministic_algorithms(False)
        print(f'{str(dtype)[6:]:9s} K={K:3d} E={E:9d}: uniform atomic {out[0]:.3f} | sorted {out[1]:.3f} ms    0.25 % hub: atomic {out[2]:.3f} | sorted {out[3]:.3f} ms',
              flush=True)
