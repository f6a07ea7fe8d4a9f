"""segment_matmul (C2) forward + backward timing, bf16 (C4: tools/c4_parts.py):
    python tools/bench_backward.py [scale]"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyg_lib_amd import ops
dev = torch.device('cuda:0')
DT = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == 'f32' else torch.bfloat16
x, ptr, w, (N, B, F) = bench.make_c2(dev, 0, 1, DT, float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
x.requires_grad_(True); w.requires_grad_(True)
gy = torch.randn(N, F, device=dev, dtype=DT)
def T(f, n=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
fwd = T(lambda: ops.segment_matmul(x.detach(), ptr, w.detach()))
def fb(gx, gw):
    a = x if gx else x.detach(); b = w if gw else w.detach()
    y = ops.segment_matmul(a, ptr, b)
    torch.autograd.grad(y, [t for t in (a, b) if t.requires_grad], gy)
r = dict(workload='C2 segment_matmul ' + str(DT), rows=N, fwd_ms=round(fwd, 3), fwd_bwd_dx_ms=round(T(lambda: fb(True, False)), 3),
         fwd_bwd_dw_ms=round(T(lambda: fb(False, True)), 3), fwd_bwd_both_ms=round(T(lambda: fb(True, True)), 3))
print(json.dumps(r))
# C4 (grouped_matmul) forward / dX / dW: tools/c4_parts.py
