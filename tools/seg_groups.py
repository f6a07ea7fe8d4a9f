"""segment_matmul kernel time (torch events) for a few (K, M, groups): python tools/seg_groups.py [K M]"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyg_lib_amd import ops
dev = 'cuda:0'
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = 6_000_000 * 256 // max(K, M)
x = torch.randn(N, K, device=dev).bfloat16()
for B in (16, 512):
    ptr = torch.linspace(0, N, B + 1).long(); ptr[-1] = N
    w = (torch.randn(B, K, M, device=dev) / 16).bfloat16()
    for _ in range(4): ops.segment_matmul(x, ptr, w)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); ops.segment_matmul(x, ptr, w); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    gb = N * (K + M) * 2 / 1e9
    print(f'K {K} M {M} B {B}: {min(ts):.3f} ms  {gb / min(ts):.2f} TB/s  {ops.matmul_last_variant()}')
