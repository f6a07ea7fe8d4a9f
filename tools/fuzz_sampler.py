"""Differential fuzzing of the HIP sampler against the oracle: random homogeneous / heterogeneous graphs,
fan-outs (incl. 0, -1, > 64), duplicate seeds, replace / disjoint / temporal / biased (edge_weight) modes, int32
graphs; every output and the generator state must match bit for bit.    python tools/fuzz_sampler.py [cases] [seed]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from pyg_lib_amd import sampler
I64_MIN, I64_MAX = -2 ** 63, 2 ** 63 - 1
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def run(cases=200, seed=0):
    rng = np.random.default_rng(seed)


    def csr(rows, cols, mean):
        deg = rng.poisson(mean, rows).astype(np.int64)
        deg[rng.random(rows) < 0.15] = 0
        if rng.random() < 0.2 and rows > 3:
            deg[rng.integers(0, rows)] = min(cols * 3, 400)  # a hub
        rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        return rp, rng.integers(0, max(cols, 1), int(rp[-1]), dtype=np.int64)


    def fanout(L):
        return [int(rng.choice([-1, 0, 1, 2, 3, 5, 8, 17, 40, 70], p=[.06, .06, .12, .14, .14, .16, .12, .1, .06, .04]))
                for _ in range(L)]


    def weights(n):
        kind = int(rng.integers(0, 4))
        if kind == 0:
            w = rng.random(n) + 0.01
        elif kind == 1:
            w = np.ones(n)
        elif kind == 2:
            w = (rng.random(n) < 0.4).astype(np.float64)  # zero weights: -inf keys, ties
        else:
            w = rng.integers(1, 3, n).astype(np.float64)
        return w.astype(np.float32 if rng.random() < 0.6 else np.float64)


    t0 = time.time()
    nh = nt = nb = 0
    for it in range(cases):
        seed = int(rng.integers(0, 2 ** 31))
        replace, disjoint = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        L = int(rng.integers(1, 4))
        if rng.random() < 0.55:  # homogeneous, possibly temporal
            n = int(rng.integers(1, 2500))
            rp, cl = csr(n, n, float(rng.choice([1.5, 6, 20])))
            seeds = rng.integers(0, n, int(rng.integers(0, 60)))
            kw = dict(replace=replace, disjoint=disjoint)
            if rng.random() < 0.35 and seeds.size:
                kw['disjoint'] = True
                kw['temporal_strategy'] = str(rng.choice(['uniform', 'last']))
                if rng.random() < 0.5:
                    nt_ = rng.integers(0, 50, n, dtype=np.int64)
                    for v in range(n):
                        a, b = rp[v], rp[v + 1]
                        cl[a:b] = cl[a:b][np.argsort(nt_[cl[a:b]], kind='stable')]
                    kw['node_time'] = nt_
                    if rng.random() < 0.5:
                        kw['seed_time'] = rng.integers(0, 60, seeds.size, dtype=np.int64)
                else:
                    et = rng.integers(0, 50, cl.size, dtype=np.int64)
                    for v in range(n):
                        et[rp[v]:rp[v + 1]] = np.sort(et[rp[v]:rp[v + 1]])
                    kw['edge_time'] = et
                    kw['seed_time'] = rng.integers(0, 60, seeds.size, dtype=np.int64)
                nt += 1
            biased = 'temporal_strategy' not in kw and rng.random() < 0.3
            fan = fanout(L)
            if biased:
                kw['edge_weight'] = weights(cl.size)
                if kw['replace']:  # at::multinomial: every sampled row needs a positive weight
                    kw['edge_weight'] = np.abs(kw['edge_weight']) + kw['edge_weight'].dtype.type(0.25)
                nb += 1
            torch.manual_seed(seed)
            dkw = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
            if not biased and rng.random() < 0.15:  # int32 graph: same samples, int32 outputs
                o32 = sampler.neighbor_sample(dev(rp).int(), dev(cl).int(), dev(seeds.astype(np.int64)).int(), fan, **dkw)
                assert all(o32[i].dtype == torch.int32 for i in range(4))
                torch.manual_seed(seed)
                o64 = sampler.neighbor_sample(dev(rp), dev(cl), dev(seeds.astype(np.int64)), fan, **dkw)
                if not all(torch.equal(o32[i].long(), o64[i]) for i in range(4)):
                    print('MISMATCH int32 vs int64 at case', it, 'seed', seed)
                    return False
                torch.manual_seed(seed)
            out = sampler.neighbor_sample(dev(rp), dev(cl), dev(seeds.astype(np.int64)), fan, **dkw)
            after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
            ref = oracle.neighbor_sample(rp, cl, seeds.astype(np.int64), fan, rng_seed=seed, **kw)
            ok = all(torch.equal(out[i].cpu(), torch.from_numpy(ref[i])) for i in range(4)) and out[4] == ref[4] and out[5] == ref[5]
            info = ref[6]
        else:
            T = int(rng.integers(2, 5))
            types = [f't{i}' for i in range(T)]
            sizes = {t: int(rng.integers(1, 1500)) for t in types}
            R = int(rng.integers(1, 8))
            ets = []
            for r in range(R):
                ets.append((types[int(rng.integers(0, T))], f'r{r}', types[int(rng.integers(0, T))]))
            rp, cl = {}, {}
            for (s_, r_, d_) in ets:
                rp[(s_, r_, d_)], cl[(s_, r_, d_)] = csr(sizes[s_], sizes[d_], float(rng.choice([2, 7])))
            seed_types = [t for t in types if rng.random() < 0.6] or [types[0]]
            seeds = {t: rng.integers(0, sizes[t], int(rng.integers(1, 40))).astype(np.int64) for t in seed_types}
            fan = {e: fanout(L) for e in ets}
            wd = None
            if rng.random() < 0.35:  # biased: all or some of the relations weighted
                some = rng.random() < 0.5
                wd = {e: weights(cl[e].size) for e in ets if not some or rng.random() < 0.5}
                if not wd:
                    wd = {ets[0]: weights(cl[ets[0]].size)}
                if replace:  # at::multinomial: every sampled row needs a positive weight
                    wd = {e: np.abs(v) + v.dtype.type(0.25) for e, v in wd.items()}
                nb += 1
            torch.manual_seed(seed)
            out = sampler.hetero_neighbor_sample({e: dev(v) for e, v in rp.items()}, {e: dev(v) for e, v in cl.items()},
                                                 {k: dev(v) for k, v in seeds.items()}, fan, replace=replace, disjoint=disjoint,
                                                 edge_weight_dict=None if wd is None else {e: dev(v) for e, v in wd.items()})
            after = int(torch.randint(I64_MIN, I64_MAX, (1,)).item())
            ref = oracle.hetero_neighbor_sample(types, ets, rp, cl, seeds, fan, replace=replace, disjoint=disjoint, rng_seed=seed,
                                                edge_weight_dict=wd)
            ok = True
            for e in ets:
                ok = ok and torch.equal(out[0][e].cpu(), torch.from_numpy(ref[0][e])) and torch.equal(out[1][e].cpu(), torch.from_numpy(ref[1][e]))
                ok = ok and torch.equal(out[3][e].cpu(), torch.from_numpy(ref[3][e])) and out[5][e] == ref[5][e]
            for t in types:
                if t in out[2]:
                    ok = ok and torch.equal(out[2][t].cpu(), torch.from_numpy(ref[2][t])) and out[4][t] == ref[4][t]
                else:  # the wrapper only knows node types named by an edge type or a seed set
                    ok = ok and ref[2][t].size == 0
            info = ref[6]
            nh += 1
        ok = ok and after == oracle.mt19937_word_after(seed, info['rng_blocks'] * 256 + info['rng_raw_draws'])
        if not ok:
            print('MISMATCH at case', it, 'seed', seed, 'replace', replace, 'disjoint', disjoint, 'L', L)
            return False
    print(f'{cases} cases ({nh} hetero, {nt} temporal, {nb} biased) match the oracle bit for bit in {time.time() - t0:.1f}s')
    return True


if __name__ == '__main__':
    ok = run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sys.exit(0 if ok else 1)
