"""C4-shaped grouped_matmul (512 variable groups, F=256, bf16) and the fp32 / fp16 variants of C2."""
import os, sys, json, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyg_lib_amd import ops

dev = torch.device('cuda:0')


def timeit(f, n=10, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n


g = torch.Generator().manual_seed(0)
rows = torch.exp(torch.rand(512, generator=g) * (math.log(65536) - math.log(256)) + math.log(256)).long().tolist()
F = 256
ins = [torch.randn(r, F, device=dev).bfloat16() for r in rows]
oth = [(torch.randn(F, F, device=dev) / 16).bfloat16() for _ in rows]
ms = timeit(lambda: ops.grouped_matmul(ins, oth))
n = sum(rows)
flops = 2.0 * n * F * F
byts = 2 * (2 * n * F + 512 * F * F)
print('C4 grouped_matmul', json.dumps(dict(rows=n, ms=round(ms, 3), TFLOPs=round(flops / ms / 1e9, 1), GBps=round(byts / ms / 1e6, 1),
                                            variant=ops.matmul_last_variant())))
ms_t = timeit(lambda: [a @ b for a, b in zip(ins, oth)], n=3, warm=1)
print('   torch per-group @ loop', round(ms_t, 3), 'ms')
del ins, oth
for dt, name in ((torch.float32, 'f32'), (torch.float16, 'f16')):
    x, ptr, w, (N, B, Fd) = bench.make_c2(dev, 0, 1, dt)
    ms = timeit(lambda: ops.segment_matmul(x, ptr, w), n=5)
    s = x.element_size()
    print(f'C2 segment_matmul {name}', json.dumps(dict(ms=round(ms, 3), TFLOPs=round(2.0 * N * Fd * Fd / ms / 1e9, 1),
                                                       GBps=round(s * 2 * N * Fd / ms / 1e6, 1), variant=ops.matmul_last_variant())))
    del x, w
