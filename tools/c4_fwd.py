"""C4 forward only (512 groups, F=256, bf16), a few launches -- for rocprofv3 PMC passes."""
import os, sys, math, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyg_lib_amd  # noqa
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
sizes = torch.exp(torch.rand(512, generator=g) * (math.log(65536.0) - math.log(256.0)) + math.log(256.0)).long().tolist()
xs = [torch.randn(n, 256, device=dev).to(torch.bfloat16) for n in sizes]
ws = [(torch.randn(256, 256, device=dev) / 16).to(torch.bfloat16) for _ in sizes]
for _ in range(2): torch.ops.pyg.grouped_matmul(xs, ws)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): torch.ops.pyg.grouped_matmul(xs, ws)
torch.cuda.synchronize()
rows = sum(sizes)
ms = (time.perf_counter() - t) / 5 * 1e3
print(f'rows {rows} fwd {ms:.3f} ms  alg bytes {rows * 256 * 2 * 2 / 1e9:.2f} GB -> {rows * 256 * 2 * 2 / ms / 1e9:.2f} TB/s')
