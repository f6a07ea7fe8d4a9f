"""State stress of the sampler's kept resources (node-table cache, random-word stream): a random sequence of calls -- batch
sizes 1 ... 30000, fan-outs, plain / disjoint / replace, two graphs of different sizes and a heterogeneous one taking turns,
reseeding, foreign draws from the generator, release_table_cache() -- every call against the oracle on the SAME torch generator
(its `fill` hook), the generator state compared at the end.   python tools/stress_sampler_state.py [calls] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from pyg_lib_amd import sampler  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def torch_fill(arr):
    arr[:] = torch.randint(-2 ** 63, 2 ** 63 - 1, (128,)).numpy()


def csr(n, deg):
    d = rng.poisson(deg, n).astype(np.int64)
    d[rng.random(n) < 0.05] = 0
    return np.concatenate([[0], np.cumsum(d)]).astype(np.int64), rng.integers(0, n, int(d.sum()), dtype=np.int64)


graphs = []
for n, deg in ((300_000, 25), (40_000, 12)):
    rp, cl = csr(n, deg)
    graphs.append((n, rp, cl, dev(rp), dev(cl)))
types = ['a', 'b']
ets = [('a', 'x', 'a'), ('a', 'y', 'b'), ('b', 'z', 'a')]
hn = {'a': 50_000, 'b': 20_000}
hrp, hcl = {}, {}
for et in ets:
    d = rng.poisson(8, hn[et[0]]).astype(np.int64)
    hrp[et] = np.concatenate([[0], np.cumsum(d)]).astype(np.int64)
    hcl[et] = rng.integers(0, hn[et[2]], int(d.sum()), dtype=np.int64)
hrpd, hcld = {e: dev(v) for e, v in hrp.items()}, {e: dev(v) for e, v in hcl.items()}

plan = []
for i in range(calls):
    r = rng.random()
    if r < 0.06:
        plan.append(('reseed', int(rng.integers(0, 2 ** 31))))
    elif r < 0.12:
        plan.append(('draw', int(rng.integers(1, 300))))
    elif r < 0.16:
        plan.append(('release',))
    elif r < 0.30:
        b = int(rng.choice([1, 7, 300, 2000]))
        plan.append(('hetero', rng.permutation(hn['a'])[:b].astype(np.int64), [int(rng.choice([2, 5, 12])), int(rng.choice([1, 4]))],
                     bool(rng.integers(0, 2))))
    else:
        gi = int(rng.integers(0, 2))
        n = graphs[gi][0]
        b = int(rng.choice([1, 3, 64, 500, 1024, 4000, 12000, 30000]))
        b = min(b, n)
        L = int(rng.integers(1, 4))
        fan = [int(rng.choice([1, 3, 5, 10, 15, 25])) for _ in range(L)]
        mode = int(rng.integers(0, 3))
        plan.append(('homo', gi, rng.permutation(n)[:b].astype(np.int64), fan, dict(disjoint=mode == 1, replace=mode == 2)))


def run(device):
    torch.manual_seed(12345)
    outs = []
    for p in plan:
        if p[0] == 'reseed':
            torch.manual_seed(p[1])
            outs.append(None)
        elif p[0] == 'draw':
            torch.rand(p[1])
            outs.append(None)
        elif p[0] == 'release':
            if device:
                sampler.release_table_cache()
            outs.append(None)
        elif p[0] == 'hetero':
            _, s, f, dis = p
            fan = {e: f for e in ets}
            if device:
                o = sampler.hetero_neighbor_sample(hrpd, hcld, {'a': dev(s)}, fan, disjoint=dis)
                outs.append([o[0], o[1], o[2]])
            else:
                o = oracle.hetero_neighbor_sample(types, ets, hrp, hcl, {'a': s}, fan, disjoint=dis, fill=torch_fill)
                outs.append([o[0], o[1], o[2]])
        else:
            _, gi, s, fan, kw = p
            n, rp, cl, rpd, cld = graphs[gi]
            if device:
                outs.append(sampler.neighbor_sample(rpd, cld, dev(s), fan, **kw)[:4])
            else:
                outs.append(oracle.neighbor_sample(rp, cl, s, fan, fill=torch_fill, **kw)[:4])
    if device:
        torch.cuda.synchronize()
    return outs, torch.get_rng_state()


t0 = time.time()
got, gs = run(True)
ref, rs = run(False)
bad = 0
for i, (g, r) in enumerate(zip(got, ref)):
    if g is None:
        continue
    if plan[i][0] == 'hetero':
        for dg, dr in zip(g, r):
            for k in dr:
                if not np.array_equal(dg[k].cpu().numpy().reshape(-1), np.asarray(dr[k]).reshape(-1)):
                    bad += 1
    else:
        for a, b in zip(g, r):
            if not np.array_equal(a.cpu().numpy().reshape(-1), np.asarray(b).reshape(-1)):
                bad += 1
ok_state = torch.equal(gs, rs)
print(f'{calls} steps (seed {seed}): {sum(p[0] in ("homo", "hetero") for p in plan)} sampler calls, {bad} mismatching outputs, '
      f'generator state {"equal" if ok_state else "DIFFERENT"}, carry stats {sampler.rng_carry_stats()}, {time.time() - t0:.1f} s')
sys.exit(1 if bad or not ok_state else 0)
