"""Generates tests/golden/biased_golden.npz: biased (edge_weight) neighbour sampling computed with the REAL
libtorch CPU ops the reference calls -- pyg_lib/csrc/sampler/cpu/neighbor_kernel.cpp:245-285:

    rand  = at::empty_like(weight).uniform_()         # global CPU generator
    key   = rand.log() / weight
    index = std::get<1>(key.topk(count))              # replace == false
    index = at::multinomial(weight, count, true)      # replace == true (count > 1 and the single-draw route)

around a Python transcription of the (single-threaded) hop loop (:332-514 homogeneous, :518-841
heterogeneous; Mapper = first-occurrence dict).  The engine constructor's prefetch
(`at::randint(INT64_MIN, INT64_MAX, {128})`, random/cpu/rand_engine.h:27-29) is drawn first, as in the
reference.  The reference kernel itself cannot be built here; these vectors pin the oracle's
restatement (oracle_sampler.c: sampler_biased, oracle_topk.cpp), which restates `log` as the correctly rounded
logarithm -- the script also counts how many of the 2^24 possible float32 inputs libtorch rounds differently.

Run from the repo root:  python tests/golden/make_biased_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
I64_MIN, I64_MAX = -2**63, 2**63 - 1


def torch_biased(node_types, edge_types, rowptr, col, weight, seed_dict, fanouts, manual_seed, disjoint=False, csc=False,
                 replace=False):
    torch.manual_seed(manual_seed)
    torch.randint(I64_MIN, I64_MAX, (128,))  # RandintEngine constructor prefetch
    nodes = {t: [] for t in node_types}       # (batch, node) or node
    mapper = {t: {} for t in node_types}
    slices = {}
    b = 0
    for t, s in seed_dict.items():
        for v in s:
            key = (b, int(v)) if disjoint else int(v)
            nodes[t].append(key)              # non-disjoint: duplicates stay in the list (mapper.fill)
            if key not in mapper[t]:
                mapper[t][key] = len(nodes[t]) - 1 if disjoint else len(mapper[t])
            b += 1
    if not disjoint:
        # Mapper::fill assigns consecutive ids to first occurrences, but sampled_nodes keeps duplicates; the
        # local id of a later-sampled node is the mapper's counter, which the reference seeds with the number
        # of DISTINCT seeds.  (Fixtures below avoid duplicate seeds so both views agree.)
        pass
    for t in node_types:
        slices[t] = (0, len(nodes[t]))
    rows = {e: [] for e in edge_types}
    cols = {e: [] for e in edge_types}
    eids = {e: [] for e in edge_types}
    nhops = {t: [len(nodes[t])] for t in node_types}
    ehops = {e: [] for e in edge_types}
    L = len(next(iter(fanouts.values())))
    for ell in range(L):
        for e in edge_types:
            src, dst = (e[0], e[2]) if not csc else (e[2], e[0])
            count = fanouts[e][ell]
            ehops[e].append(0)
            lo, hi = slices[src]
            rp, cl, w = rowptr[e], col[e], weight[e]
            for i in range(lo, hi):
                key = nodes[src][i]
                v = key[1] if disjoint else key
                rs, re = int(rp[v]), int(rp[v + 1])
                if re - rs == 0 or count == 0:
                    continue
                if count < 0 or (not replace and count >= re - rs):
                    picked = list(range(rs, re))
                elif replace:
                    # neighbor_kernel.cpp:267-270 (count > 1: the with-replacement kernel of at::multinomial)
                    picked = (rs + torch.multinomial(w[rs:re], count, True)).tolist()
                else:
                    ww = w[rs:re]
                    rand = torch.empty_like(ww).uniform_()
                    k = rand.log() / ww
                    picked = (rs + k.topk(count)[1]).tolist()
                for ed in picked:
                    u = int(cl[ed])
                    dk = (key[0], u) if disjoint else u
                    if dk not in mapper[dst]:
                        mapper[dst][dk] = len(nodes[dst])
                        nodes[dst].append(dk)
                    rows[e].append(i)
                    cols[e].append(mapper[dst][dk])
                    eids[e].append(ed)
                    ehops[e][-1] += 1
        for t in node_types:
            slices[t] = (slices[t][1], len(nodes[t]))
            nhops[t].append(slices[t][1] - slices[t][0])
    return rows, cols, nodes, eids, nhops, ehops


def random_csr(g, n_src, n_dst, avg_deg, hub=0):
    deg = torch.poisson(torch.full((n_src,), float(avg_deg)), generator=g).long()
    if hub:
        deg[torch.randint(0, n_src, (2,), generator=g)] = hub
    rowptr = torch.zeros(n_src + 1, dtype=torch.long)
    rowptr[1:] = deg.cumsum(0)
    col = torch.randint(0, n_dst, (int(rowptr[-1]),), generator=g)
    return rowptr, col


def make_weight(g, n, kind, dtype):
    if kind == 'random':
        w = torch.rand(n, generator=g, dtype=torch.float64) + 0.05
    elif kind == 'ones':
        w = torch.ones(n, dtype=torch.float64)
    elif kind == 'zeros_mixed':      # many zero weights: -inf keys, ties among them
        w = (torch.rand(n, generator=g, dtype=torch.float64) < 0.35).double()
    else:
        w = torch.randint(1, 4, (n,), generator=g).double()  # few distinct weights
    return w.to(dtype)


def main():
    import oracle
    # how far is libtorch's float log from the correctly rounded one on the uniform_ domain?
    u = (torch.arange(0, 1 << 24, dtype=torch.int64).double() * 2.0**-24).float()
    t_log = torch.log(u)
    cr = torch.log(u.double()).float()
    print('float32 inputs k*2^-24 where libtorch log != correctly rounded log:', int((t_log != cr).sum()), 'of', 1 << 24)

    g = torch.Generator().manual_seed(20260926)
    out = {}
    cases = []
    n_mismatch = 0
    n_rows = 0
    cid = 0
    for kind in ('random', 'ones', 'zeros_mixed', 'small_ints'):
        for dtype in (torch.float32, torch.float64):
            for disjoint in (False, True):
                for hetero in (False, True):
                    for trial in range(2):
                        ms = 1000 + cid
                        if not hetero:
                            nt, ets = ['n'], [('n', 'to', 'n')]
                            n = 300
                            rp, cl = random_csr(g, n, n, 12, hub=400 if trial else 0)
                            rowptr, col = {ets[0]: rp}, {ets[0]: cl}
                            fan = {ets[0]: [4, 3] if trial == 0 else [5, 2, 2]}
                            seeds = {'n': torch.randperm(n, generator=g)[:7]}
                        else:
                            nt = ['a', 'b']
                            ets = [('a', 'x', 'b'), ('b', 'y', 'a'), ('a', 'z', 'a')]
                            sizes = {'a': 200, 'b': 120}
                            rowptr, col = {}, {}
                            for e in ets:
                                rowptr[e], col[e] = random_csr(g, sizes[e[0]], sizes[e[2]], 9, hub=700 if trial else 0)
                            fan = {ets[0]: [3, 2], ets[1]: [2, 4], ets[2]: [6, 1]}
                            seeds = {'a': torch.randperm(200, generator=g)[:5], 'b': torch.randperm(120, generator=g)[:3]}
                        weight = {e: make_weight(g, col[e].numel(), kind, dtype) for e in ets}
                        ref = torch_biased(nt, ets, rowptr, col, weight, seeds, fan, ms, disjoint=disjoint)
                        got = oracle.hetero_neighbor_sample(
                            nt, ets, {e: rowptr[e].numpy() for e in ets}, {e: col[e].numpy() for e in ets},
                            {t: s.numpy() for t, s in seeds.items()}, fan, disjoint=disjoint, rng_seed=ms,
                            edge_weight_dict={e: weight[e].numpy() for e in ets})
                        ok = True
                        for e in ets:
                            ok &= got[0][e].tolist() == ref[0][e] and got[1][e].tolist() == ref[1][e]
                            ok &= got[3][e].tolist() == ref[3][e] and got[5][e] == ref[5][e]
                        for t in nt:
                            ok &= got[2][t].tolist() == [list(x) if disjoint else x for x in ref[2][t]]
                            ok &= got[4][t] == ref[4][t]
                        n_rows += sum(len(v) for v in ref[0].values())
                        if not ok:
                            n_mismatch += 1
                            print('oracle != torch transcription:', kind, dtype, disjoint, hetero, trial)
                        pre = f'c{cid}_'
                        out[pre + 'meta'] = np.array([int(hetero), int(disjoint), int(dtype == torch.float64), ms, len(ets)])
                        for j, e in enumerate(ets):
                            out[pre + f'rowptr{j}'] = rowptr[e].numpy()
                            out[pre + f'col{j}'] = col[e].numpy()
                            out[pre + f'weight{j}'] = weight[e].numpy()
                            out[pre + f'fan{j}'] = np.array(fan[e])
                            out[pre + f'row_out{j}'] = np.array(ref[0][e], dtype=np.int64)
                            out[pre + f'col_out{j}'] = np.array(ref[1][e], dtype=np.int64)
                            out[pre + f'edge_out{j}'] = np.array(ref[3][e], dtype=np.int64)
                            out[pre + f'ehops{j}'] = np.array(ref[5][e], dtype=np.int64)
                        for t in nt:
                            out[pre + f'seed_{t}'] = seeds[t].numpy()
                            out[pre + f'node_{t}'] = np.array(ref[2][t], dtype=np.int64)
                            out[pre + f'nhops_{t}'] = np.array(ref[4][t], dtype=np.int64)
                        cases.append((kind, str(dtype), disjoint, hetero, trial))
                        cid += 1
    # with replacement (at::multinomial, count > 1): strictly positive weights (an all-zero row is an error)
    for kind in ('random', 'small_ints'):
        for dtype in (torch.float32, torch.float64):
            for hetero in (False, True):
                for disjoint in (False, True):
                    ms = 1000 + cid
                    if not hetero:
                        nt, ets = ['n'], [('n', 'to', 'n')]
                        n = 300
                        rp, cl = random_csr(g, n, n, 9, hub=300)
                        rowptr, col = {ets[0]: rp}, {ets[0]: cl}
                        fan = {ets[0]: [4, 1, 2]}  # a fan-out of 1 takes at::multinomial's single-draw route
                        seeds = {'n': torch.randperm(n, generator=g)[:7]}
                    else:
                        nt = ['a', 'b']
                        ets = [('a', 'x', 'b'), ('b', 'y', 'a'), ('a', 'z', 'a')]
                        sizes = {'a': 200, 'b': 120}
                        rowptr, col = {}, {}
                        for e in ets:
                            rowptr[e], col[e] = random_csr(g, sizes[e[0]], sizes[e[2]], 7, hub=500)
                        fan = {ets[0]: [3, 1], ets[1]: [1, 4], ets[2]: [6, 2]}
                        seeds = {'a': torch.randperm(200, generator=g)[:5], 'b': torch.randperm(120, generator=g)[:3]}
                    weight = {e: make_weight(g, col[e].numel(), kind, dtype) for e in ets}
                    ref = torch_biased(nt, ets, rowptr, col, weight, seeds, fan, ms, disjoint=disjoint, replace=True)
                    got = oracle.hetero_neighbor_sample(
                        nt, ets, {e: rowptr[e].numpy() for e in ets}, {e: col[e].numpy() for e in ets},
                        {t: s.numpy() for t, s in seeds.items()}, fan, disjoint=disjoint, rng_seed=ms, replace=True,
                        edge_weight_dict={e: weight[e].numpy() for e in ets})
                    ok = all(got[3][e].tolist() == ref[3][e] and got[1][e].tolist() == ref[1][e] for e in ets)
                    n_rows += sum(len(v) for v in ref[0].values())
                    if not ok:
                        n_mismatch += 1
                        print('oracle != torch transcription (replace):', kind, dtype, disjoint, hetero)
                    pre = f'c{cid}_'
                    out[pre + 'meta'] = np.array([int(hetero), int(disjoint), int(dtype == torch.float64), ms, len(ets), 1])
                    for j, e in enumerate(ets):
                        out[pre + f'rowptr{j}'] = rowptr[e].numpy()
                        out[pre + f'col{j}'] = col[e].numpy()
                        out[pre + f'weight{j}'] = weight[e].numpy()
                        out[pre + f'fan{j}'] = np.array(fan[e])
                        out[pre + f'row_out{j}'] = np.array(ref[0][e], dtype=np.int64)
                        out[pre + f'col_out{j}'] = np.array(ref[1][e], dtype=np.int64)
                        out[pre + f'edge_out{j}'] = np.array(ref[3][e], dtype=np.int64)
                        out[pre + f'ehops{j}'] = np.array(ref[5][e], dtype=np.int64)
                    for t in nt:
                        out[pre + f'seed_{t}'] = seeds[t].numpy()
                        out[pre + f'node_{t}'] = np.array(ref[2][t], dtype=np.int64)
                        out[pre + f'nhops_{t}'] = np.array(ref[4][t], dtype=np.int64)
                    cid += 1
    out['num_cases'] = np.array([cid])
    np.savez_compressed(os.path.join(HERE, 'biased_golden.npz'), **out)
    print(f'{cid} cases, {n_rows} sampled edges, oracle mismatches: {n_mismatch}')


if __name__ == '__main__':
    main()
