"""Many batched sampler calls back to back (C3 homogeneous K = 16, C5 hetero K = 8): the lanes' host threads, the table
cache (one table per lane in flight) and the one-launch scans' look-back under overlap must never hang, and every batch
of a repeated call must stay identical to the first call's.   python tools/stress_sampler_batched.py [calls]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler, bench_legs
from pyg_lib_amd import sampler
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
rowptr, col = bench_sampler.make_graph(dev)
rp, cl = bench_legs.make_mag_graph(dev)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
fan = {e: [15, 10] for e in ets}
g = torch.Generator().manual_seed(7)
K, KH = 16, 8
seeds = [s.to(dev) for s in torch.randperm(bench_sampler.N_NODES, generator=g)[:1024 * K].view(K, -1)]
pseeds = [{'paper': s.to(dev)} for s in torch.randperm(bench_legs.MAG_SIZES['paper'], generator=g)[:1024 * KH].view(KH, -1)]
gs, gh = list(range(100, 100 + K)), list(range(500, 500 + KH))
ref = sampler.neighbor_sample_batched(rowptr, col, seeds, [15, 10, 5], gs)
href = sampler.hetero_neighbor_sample_batched(rp, cl, pseeds, fan, gh)
# every batch against the single-batch operator once
for b in (0, K - 1):
    torch.manual_seed(gs[b])
    one = sampler.neighbor_sample(rowptr, col, seeds[b], [15, 10, 5])
    assert all(torch.equal(one[i], ref[b][i]) for i in range(4)) and one[4] == ref[b][4] and one[5] == ref[b][5]
t = time.time()
bad = 0
for i in range(n):
    out = sampler.neighbor_sample_batched(rowptr, col, seeds, [15, 10, 5], gs)
    if i % 10 == 0:
        bad += sum(int(not (torch.equal(o[0], r[0]) and torch.equal(o[1], r[1]) and torch.equal(o[2], r[2]) and o[5] == r[5]))
                   for o, r in zip(out, ref))
    if i % 3 == 0:
        h = sampler.hetero_neighbor_sample_batched(rp, cl, pseeds, fan, gh)
        if i % 30 == 0:
            bad += sum(int(not all(torch.equal(a[2][k], b[2][k]) for k in b[2])) for a, b in zip(h, href))
    if i % 50 == 25:
        sampler.release_table_cache()      # tables come and go under the running loop
torch.cuda.synchronize()
print('batched calls', n, '(K = 16) + hetero', (n + 2) // 3, '(K = 8): mismatching batches', bad, 'seconds %.1f' % (time.time() - t))
