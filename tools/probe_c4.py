"""C4-shaped grouped_matmul (512 groups, rows log-uniform [256, 65536], K = M = 256, bf16): kernel variants side by side."""
import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyg_lib_amd import ops
from bench_legs import _kernel_ms
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
B = int(os.environ.get('C4_GROUPS', 512))
rows = [int(math.exp(v)) for v in (torch.rand(B, generator=g) * (math.log(65536) - math.log(256)) + math.log(256)).tolist()]
xs = [torch.randn(r, 256, device=dev).bfloat16() for r in rows]
ws = [(torch.randn(256, 256, device=dev) / 16).bfloat16() for _ in rows]
bytes_alg = sum(r * 512 * 2 for r in rows) + B * 256 * 256 * 2
ref = None
for sched in ('auto', 'cyclic', 'contiguous'):
    ops.set_matmul_schedule(sched)
    for _ in range(2):
        outs = ops.grouped_matmul(xs, ws)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        outs = ops.grouped_matmul(xs, ws)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    name = ops.matmul_last_variant()
    if ref is None:
        ref = outs
        bad = 0
        for i in (0, 1, 7, 100, B - 1):
            want = xs[i].double() @ ws[i].double()
            err = (outs[i].double() - want).abs().max().item()
            bad = max(bad, err / max(want.abs().max().item(), 1e-9))
        print('max rel err vs float64 (5 groups): %.3e' % bad)
    else:
        same = all(torch.equal(o.view(torch.int16), r.view(torch.int16)) for o, r in zip(outs, ref))
        print('bitwise equal to first variant:', same)
    kms = _kernel_ms(lambda: ops.grouped_matmul(xs, ws), iters=8, warmup=1)
    print(sched, name, '%.3f ms operator' % ms, '%.3f ms kernel' % kms, '%.2f TB/s = %.3f of 8 TB/s' % (bytes_alg / kms / 1e9, bytes_alg / kms / 8e9))
ops.set_matmul_schedule('auto')

# duty-cycle dependence: single launches separated by idle gaps vs back-to-back
import time
for sched in ('auto', 'cyclic'):
    ops.set_matmul_schedule(sched)
    ops.grouped_matmul(xs, ws); torch.cuda.synchronize()
    for gap in (0.0, 0.02, 0.2):
        v = []
        for _ in range(6):
            time.sleep(gap)
            v.append(_kernel_ms(lambda: ops.grouped_matmul(xs, ws), iters=1, warmup=0))
        print(sched, ops.matmul_last_variant(), 'idle gap %.2f s: kernel ms' % gap, ' '.join('%.3f' % t for t in v))
ops.set_matmul_schedule('auto')
