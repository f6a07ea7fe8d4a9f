"""Stress / leak check: many sampler and reduce calls of random shapes, allocator usage must return to the
starting level and random spot checks must match the oracle.   python tools/stress.py [iters]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
import bench_sampler
from pyg_lib_amd import ops, sampler
dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rowptr, col = bench_sampler.make_graph(dev)
rp_np, cl_np = rowptr.cpu().numpy(), col.cpu().numpy()
rng = np.random.default_rng(0)
torch.cuda.synchronize()
base = torch.cuda.memory_allocated()
t0 = time.time()
checked = 0
for it in range(iters):
    S = int(rng.integers(1, 3000))
    L = int(rng.integers(1, 4))
    fan = [int(rng.choice([-1, 1, 2, 5, 10, 15, 40, 70])) for _ in range(L)]
    if -1 in fan or 70 in fan:
        fan = fan[:2]  # keep unbounded hops small
        S = min(S, 64)
    seeds = rng.integers(0, bench_sampler.N_NODES, S)
    kw = dict(replace=bool(rng.integers(0, 2)), disjoint=bool(rng.integers(0, 2)))
    torch.manual_seed(it)
    out = sampler.neighbor_sample(rowptr, col, torch.from_numpy(seeds).to(dev), fan, **kw)
    if it % 25 == 0:
        ref = oracle.neighbor_sample(rp_np, cl_np, seeds.astype(np.int64), fan, rng_seed=it, **kw)
        assert torch.equal(out[0].cpu(), torch.from_numpy(ref[0])) and torch.equal(out[1].cpu(), torch.from_numpy(ref[1]))
        assert torch.equal(out[2].cpu(), torch.from_numpy(ref[2])) and torch.equal(out[3].cpu(), torch.from_numpy(ref[3]))
        checked += 1
    del out
    n = int(rng.integers(1, 3_000_000))
    keys = torch.randint(0, int(rng.integers(1, 2 ** 40)), (n,), device=dev)
    v, i = ops.index_sort(keys)
    if it % 25 == 0:
        tv, ti = torch.sort(keys, stable=True)
        assert torch.equal(v, tv) and torch.equal(i, ti)
    E, N, K = int(rng.integers(1, 200_000)), int(rng.integers(1, 5000)), int(rng.choice([1, 3, 8, 64, 128]))
    src = torch.randn(E, K, device=dev)
    idx = torch.randint(0, N, (E,), device=dev)
    a = ops.scatter_sum(src, idx, 0, None, N)
    b, arg = ops.scatter_max(src, idx, 0, None, N)
    sidx = torch.sort(idx).values
    c = ops.segment_sum_coo(src, sidx, None, N)
    if it % 25 == 0:
        torch.testing.assert_close(a, torch.zeros(N, K, device=dev).index_add_(0, idx, src), rtol=1e-4, atol=1e-3)
        ref = torch.full((N, K), float('-inf'), device=dev).scatter_reduce_(0, idx[:, None].expand(E, K), src, 'amax')
        ref[ref == float('-inf')] = 0
        assert torch.equal(b, ref)
        torch.testing.assert_close(c, torch.zeros(N, K, device=dev).index_add_(0, sidx, src), rtol=1e-4, atol=1e-3)
    del keys, v, i, src, idx, a, b, arg, sidx, c
    tv = ti = ref = None
torch.cuda.synchronize()
kept = torch.cuda.memory_allocated() - base   # the node table and the random-word stream the sampler keeps between calls (by design)
sampler.release_table_cache()
leak = torch.cuda.memory_allocated() - base
print(f'kept between calls: {kept} bytes; after sampler.release_table_cache(): {leak}')
print(f'{iters} iterations in {time.time() - t0:.1f}s, {checked} oracle checks, allocator delta {leak} bytes')
assert leak == 0
