"""Micro-benchmarks of the byte-bound ops against their algorithmic HBM floors."""
import os, sys, time, json
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
# index_sort: n int64 keys below 2,449,029 (3 passes)
for n in (10_000_000, 100_000_000):
    keys = torch.randint(0, 2_449_029, (n,), device=dev, generator=g)
    ms = timeit(lambda: ops.index_sort(keys, 2_449_029), n=5)
    ms_t = timeit(lambda: torch.sort(keys, stable=True), n=3, warm=1)
    floor = 8 * n + 16 * n
    passes = 3 * (8 * n + 16 * n + 16 * n)
    res[f'index_sort_n{n}'] = dict(ms=round(ms, 3), torch_sort_ms=round(ms_t, 3), Mkeys_per_s=round(n / ms / 1e3, 1),
                                   floor_GBps=round(floor / ms / 1e6, 1), lsd_budget_GBps=round(passes / ms / 1e6, 1))
    del keys
# scatter_sum / gather_coo, R-GCN aggregation shape and a big one
for (E, N, K, dt) in ((700_000, 120_000, 128, torch.bfloat16), (20_000_000, 2_000_000, 128, torch.bfloat16),
                      (20_000_000, 2_000_000, 128, torch.float32)):
    s = 2 if dt == torch.bfloat16 else 4
    src = torch.randn(E, K, device=dev, generator=g).to(dt)
    index = torch.randint(0, N, (E,), device=dev, generator=g)
    sidx = torch.sort(index).values
    out = torch.zeros(N, K, device=dev, dtype=dt)
    tag = f'E{E}_K{K}_{str(dt).split(".")[-1]}'
    bytes_sc = 8 * E + s * E * K + s * N * K
    ms = timeit(lambda: ops.scatter_sum(src, index, 0, out))
    res['scatter_sum_' + tag] = dict(ms=round(ms, 3), GBps=round(bytes_sc / ms / 1e6, 1))
    ms = timeit(lambda: ops.segment_sum_coo(src, sidx, out))
    res['segment_sum_coo_' + tag] = dict(ms=round(ms, 3), GBps=round(bytes_sc / ms / 1e6, 1))
    ms = timeit(lambda: out.index_add_(0, index, src))
    res['torch_index_add_' + tag] = dict(ms=round(ms, 3), GBps=round(bytes_sc / ms / 1e6, 1))
    feat = torch.randn(N, K, device=dev, generator=g).to(dt)
    bytes_g = 8 * E + 2 * s * E * K
    ms = timeit(lambda: ops.gather_coo(feat, sidx))
    res['gather_coo_sorted_' + tag] = dict(ms=round(ms, 3), GBps=round(bytes_g / ms / 1e6, 1))
    ms = timeit(lambda: ops.gather_coo(feat, index))
    res['gather_coo_random_' + tag] = dict(ms=round(ms, 3), GBps=round(bytes_g / ms / 1e6, 1))
    ms = timeit(lambda: ops.scatter_max(src, index, 0, None, N))
    res['scatter_max_' + tag] = dict(ms=round(ms, 3), GBps=round(bytes_sc / ms / 1e6, 1))
    del src, index, sidx, out, feat
for k, v in res.items():
    print(k, json.dumps(v))
