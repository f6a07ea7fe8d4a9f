"""C4 (512 groups, F=256, bf16) grouped_matmul: forward / dX / dW ops and the autograd round trip.
    python tools/c4_parts.py"""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyg_lib_amd import ops
dev = torch.device('cuda:0')
def T(f, n=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
g = torch.Generator().manual_seed(0)
import math
sizes = torch.exp(torch.rand(512, generator=g) * (math.log(65536.0) - math.log(256.0)) + math.log(256.0)).long().tolist()
xs = [torch.randn(n, 256, device=dev).to(torch.bfloat16) for n in sizes]
ws = [(torch.randn(256, 256, device=dev) / 16).to(torch.bfloat16) for _ in sizes]
gs = [torch.randn(n, 256, device=dev).to(torch.bfloat16) for n in sizes]
xt = [x.t() for x in xs]; wt = [w.t() for w in ws]
print('fwd op', T(lambda: torch.ops.pyg.grouped_matmul(xs, ws)))
print('dX op', T(lambda: torch.ops.pyg.grouped_matmul(gs, wt)))
print('dW op', T(lambda: torch.ops.pyg.grouped_matmul(xt, gs)))
print('list comps', T(lambda: ([x.t() for x in xs], [g_.contiguous() for g_ in gs])))
xr = [x.clone().requires_grad_(True) for x in xs]; wr = [w.clone().requires_grad_(True) for w in ws]
print('fwd via autograd fn', T(lambda: ops.grouped_matmul(xr, wr)))
def fb():
    outs = ops.grouped_matmul(xr, wr); torch.autograd.grad(outs, xr + wr, gs)
print('fwd+bwd', T(fb))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); fb(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumtime').print_stats(12)
