"""Long random-configuration run of the atomic-free fused layer against float64 (integer data: exact; every (dtype, K, M)
the kernels take; x or global tables; 1 ... 40 relations; rows of 0 ... 200 edges):  python tools/stress_rgcn_grouped.py [cases] [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyg_lib_amd import rgcn  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator().manual_seed(seed)
shapes = [(torch.bfloat16, 128, 128), (torch.float16, 128, 128), (torch.bfloat16, 256, 256), (torch.float16, 128, 256),
          (torch.bfloat16, 256, 128), (torch.float32, 128, 128), (torch.bfloat16, 64, 64), (torch.float16, 128, 64),
          (torch.bfloat16, 64, 128), (torch.float16, 40, 216), (torch.bfloat16, 192, 24), (torch.float16, 256, 16),
          (torch.bfloat16, 8, 8), (torch.bfloat16, 104, 104), (torch.float32, 64, 64), (torch.float32, 20, 128), (torch.float32, 128, 36)]


def ri(lo, hi):
    return int(torch.randint(lo, hi, (1,), generator=g))


bad = skipped = 0
for case in range(cases):
    dtype, K, M = shapes[case % len(shapes)]
    T = ri(1, 5)
    types = [f't{i}' for i in range(T)]
    n = {t: ri(1, 2000) for t in types}
    R = ri(1, 41) if case % 7 == 0 else ri(1, 9)
    ets, rows, cols = [], {}, {}
    for i in range(R):
        s, d = types[ri(0, T)], types[ri(0, T)]
        et = (s, f'r{i}', d)
        ets.append(et)
        c = [0, ri(1, 20), ri(20, 400), ri(400, 6000)][ri(0, 4)]
        hi = max(1, ri(1, n[s] + 1))
        r = torch.sort(torch.randint(0, hi, (c,), generator=g)).values
        if c and ri(0, 3) == 0:
            r[-1] = n[s] - 1
            r = torch.sort(r).values
        rows[et] = r.cuda()
        cols[et] = torch.randint(0, n[d], (c,), generator=g).cuda()
    off = rgcn.type_offsets(n, types)
    x = {t: torch.randint(-1, 2, (n[t], K), generator=g).float() for t in types}
    W = torch.zeros(R, K, M)
    k = torch.randint(0, K, (R, M), generator=g)
    W[torch.arange(R)[:, None], k, torch.arange(M)[None, :]] = (torch.randint(0, 2, (R, M), generator=g) * 2 - 1).float()
    xc = torch.cat([x[t] for t in types])
    want = torch.zeros(off['__total__'], M, dtype=torch.float64)
    for i, (s, _, d) in enumerate(ets):
        want.index_add_(0, rows[ets[i]].cpu() + off[s], xc[cols[ets[i]].cpu() + off[d]].double() @ W[i].double())
    if want.abs().max() > 256:
        skipped += 1
        continue
    if case % 2:
        y = rgcn.rgcn_layer_fused(xc.to(dtype).cuda(), off, rows, cols, ets, W.to(dtype).cuda(), grouped=True)
    else:   # through global tables and node ids
        n_glob = {t: n[t] + ri(0, 500) for t in types}
        nid = {t: torch.randperm(n_glob[t], generator=g)[:n[t]] for t in types}
        tab = {}
        for t in types:
            tab[t] = torch.randint(-3, 4, (n_glob[t], K), generator=g).float()
            tab[t][nid[t]] = x[t]
        y = rgcn.rgcn_layer_fused_tables({t: tab[t].to(dtype).cuda() for t in types}, {t: nid[t].cuda() for t in types}, types,
                                         rows, cols, ets, W.to(dtype).cuda(), grouped=True)
    if not torch.equal(y.double().cpu(), want):
        bad += 1
        print('MISMATCH case', case, dtype, K, M, n, [(e, rows[e].numel()) for e in ets], flush=True)
torch.cuda.synchronize()
print(f'{cases} cases (seed {seed}): {bad} mismatches, {skipped} skipped (sums beyond 256), pending index error {rgcn.pending_index_error()}')
sys.exit(1 if bad else 0)
