"""Micro-benchmarks of the CSR family against their algorithmic HBM floors:
segment_{sum,max}_csr / gather_csr (bytes = 8(R+1) + s E K read + s R K write) and softmax_csr
(3 reads + 2 writes of the values in the reference's three-pass form; floor = 1 read + 1 write)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyg_lib_amd import ops

dev = torch.device('cuda:0')


def timeit(f, n=10, warm=2):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n


res = {}
g = torch.Generator(device=dev).manual_seed(0)
for (E, R, K, dt) in ((20_000_000, 2_000_000, 128, torch.bfloat16), (20_000_000, 2_000_000, 128, torch.float32),
                      (100_000_000, 2_449_029, 1, torch.float32), (20_000_000, 400, 16, torch.float32)):
    s = 2 if dt == torch.bfloat16 else 4
    index = torch.sort(torch.randint(0, R, (E,), device=dev, generator=g)).values
    indptr = torch.zeros(R + 1, dtype=torch.long, device=dev)
    indptr[1:] = torch.bincount(index, minlength=R).cumsum(0)
    src = (torch.randn(E, K, device=dev, generator=g) if K > 1 else torch.randn(E, device=dev, generator=g)).to(dt)
    tag = f'E{E}_R{R}_K{K}_{str(dt).split(".")[-1]}'
    b = 8 * (R + 1) + s * E * K + s * R * K
    ms = timeit(lambda: ops.segment_sum_csr(src, indptr))
    res['segment_sum_csr_' + tag] = dict(ms=round(ms, 3), GBps=round(b / ms / 1e6, 1))
    ms = timeit(lambda: ops.segment_max_csr(src, indptr))
    res['segment_max_csr_' + tag] = dict(ms=round(ms, 3), GBps=round((b + 8 * R * K) / ms / 1e6, 1))
    ms = timeit(lambda: ops.segment_sum_coo(src, index, None, R))
    res['segment_sum_coo_' + tag] = dict(ms=round(ms, 3), GBps=round((b + 8 * E) / ms / 1e6, 1))
    red = ops.segment_sum_csr(src, indptr)
    outbuf = torch.empty_like(src)
    ms = timeit(lambda: ops.gather_csr(red, indptr, outbuf))
    res['gather_csr_' + tag] = dict(ms=round(ms, 3), GBps=round(b / ms / 1e6, 1))
    if dt == torch.float32:
        ms = timeit(lambda: ops.softmax_csr(src, indptr, 0))
        res['softmax_csr_' + tag] = dict(ms=round(ms, 3), floor_GBps=round(2 * s * E * K / ms / 1e6, 1),
                                         threepass_GBps=round(5 * s * E * K / ms / 1e6, 1))
    del src, index, indptr, red, outbuf
for k, v in res.items():
    print(k, json.dumps(v))
