"""Generates tests/golden/reduce_golden.npz with the REAL reference (build container only).

    bash oracle/build_ref.sh && python tests/golden/make_ref_golden.py

Loads oracle/_ref/libpyg_ref.so -- the reference's own index_sort / scatter / segment_coo CPU
kernels compiled unmodified from /root/reference by oracle/build_ref.sh -- and records, for a
battery of inputs, what `torch.ops.pyg.*` returns on CPU tensors.  Inputs and outputs are stored
(bf16 as uint16 bit patterns), so the tests never need the reference again.
"""
import os.path as osp

import numpy as np
import torch

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(osp.dirname(HERE))
torch.ops.load_library(osp.join(ROOT, 'oracle', '_ref', 'libpyg_ref.so'))
P = torch.ops.pyg

D = {}
META = []


def store(name, t):
    if t is None:
        return
    if isinstance(t, torch.Tensor):
        if t.dtype == torch.bfloat16:
            D[name] = t.contiguous().view(torch.int16).numpy().view(np.uint16)
            D[name + '__bf16'] = np.array(1)
        else:
            D[name] = t.contiguous().numpy()
    else:
        D[name] = np.asarray(t)


def rand(shape, dtype, g):
    if dtype in (torch.int32, torch.int64, torch.int16, torch.int8, torch.uint8):
        return torch.randint(0 if dtype == torch.uint8 else -9, 10, shape, generator=g).to(dtype)
    return torch.randn(shape, generator=g).to(dtype)


def scatter_cases():
    g = torch.Generator().manual_seed(0)
    idx8 = torch.tensor([0, 1, 0, 1, 1, 3, 2, 0])
    cases = []
    for dt in (torch.float32, torch.float64, torch.int64, torch.int32, torch.bfloat16, torch.float16):
        cases.append(dict(src=rand((8, 4), dt, g), index=idx8, dim=0))
    cases.append(dict(src=rand((3, 8), torch.float32, g), index=idx8, dim=-1))
    cases.append(dict(src=rand((3, 6, 5), torch.float32, g), index=torch.tensor([0, 2, 1, 0, 2, 1]), dim=1))
    cases.append(dict(src=rand((6, 4), torch.float32, g), index=torch.tensor([0, 1, 0, 2, 1, 2]), dim=0, dim_size=5))
    cases.append(dict(src=rand((6, 4), torch.float32, g), index=torch.tensor([0, 1, 0, 2, 1, 2]), dim=0,
                      out=rand((4, 4), torch.float32, g)))
    # full (non-broadcast) index with a K dimension
    cases.append(dict(src=rand((2, 5, 3), torch.float32, g), index=torch.randint(0, 4, (2, 5, 3), generator=g), dim=1))
    # ties for min/max (first match), negative values, empty buckets
    cases.append(dict(src=torch.tensor([[1., 1.], [1., -2.], [3., -2.], [1., 5.]]), index=torch.tensor([2, 0, 2, 0]),
                      dim=0, dim_size=4))
    cases.append(dict(src=torch.tensor([5, 5, 5, 7, 7, 1], dtype=torch.int64), index=torch.tensor([1, 1, 1, 1, 3, 3]),
                      dim=0, dim_size=5))
    # out= whose initial value ties with a source value (arg must stay the sentinel)
    cases.append(dict(src=torch.tensor([2., 3., 2.]), index=torch.tensor([0, 0, 1]), dim=0,
                      out=torch.tensor([2., 9., -1.])))
    # larger random
    cases.append(dict(src=rand((2000, 16), torch.float32, g), index=torch.randint(0, 300, (2000,), generator=g), dim=0))
    cases.append(dict(src=rand((2000, 16), torch.bfloat16, g), index=torch.randint(0, 300, (2000,), generator=g), dim=0))
    cases.append(dict(src=rand((0, 4), torch.float32, g), index=torch.zeros(0, dtype=torch.long), dim=0, dim_size=3))
    for ci, c in enumerate(cases):
        for op in ('sum', 'mul', 'mean', 'min', 'max'):
            src, index, dim = c['src'], c['index'], c['dim']
            if op == 'mean' and src.dtype in (torch.bfloat16, torch.float16) and src.numel() > 1000:
                continue
            out = c['out'].clone() if 'out' in c else None
            fn = getattr(P, 'scatter_' + op)
            res = fn(src, index, dim, out, c.get('dim_size'))
            key = f'scatter{ci}_{op}'
            base = f'scatter{ci}'
            store(base + '_src', src)
            store(base + '_index', index)
            store(base + '_dim', dim)
            if 'out' in c:
                store(base + '_out0', c['out'])
            if 'dim_size' in c:
                store(base + '_dim_size', c['dim_size'])
            if op in ('min', 'max'):
                store(key + '_res', res[0])
                store(key + '_arg', res[1])
            else:
                store(key + '_res', res)
            META.append(key)


def coo_cases():
    g = torch.Generator().manual_seed(1)
    cases = []
    for dt in (torch.float32, torch.bfloat16, torch.int64):
        cases.append(dict(src=rand((8, 4), dt, g), index=torch.tensor([0, 0, 1, 1, 1, 3, 3, 5])))
    cases.append(dict(src=rand((8,), torch.float32, g), index=torch.tensor([0, 0, 1, 1, 1, 3, 3, 5])))
    cases.append(dict(src=rand((3, 6, 2), torch.float32, g),
                      index=torch.tensor([[0, 0, 1, 1, 2, 2], [0, 1, 1, 1, 1, 3], [2, 2, 2, 2, 2, 2]])))
    cases.append(dict(src=rand((3, 6, 2), torch.float32, g), index=torch.tensor([[0, 0, 1, 1, 2, 2]])))  # [1, E] broadcast over B
    cases.append(dict(src=rand((8, 4), torch.float32, g), index=torch.tensor([0, 0, 1, 1, 1, 3, 3, 5]), dim_size=9))
    cases.append(dict(src=rand((8, 4), torch.float32, g), index=torch.tensor([0, 0, 1, 1, 1, 3, 3, 5]),
                      out=rand((7, 4), torch.float32, g)))
    idx = torch.sort(torch.randint(0, 400, (3000,), generator=g)).values
    cases.append(dict(src=rand((3000, 32), torch.float32, g), index=idx))
    cases.append(dict(src=rand((3000, 32), torch.bfloat16, g), index=idx))
    for ci, c in enumerate(cases):
        for op in ('sum', 'mean', 'min', 'max'):
            src, index = c['src'], c['index']
            if op == 'mean' and not src.is_floating_point():
                continue
            out = c['out'].clone() if 'out' in c else None
            res = getattr(P, f'segment_{op}_coo')(src, index, out, c.get('dim_size'))
            key = f'coo{ci}_{op}'
            base = f'coo{ci}'
            store(base + '_src', src)
            store(base + '_index', index)
            if 'out' in c:
                store(base + '_out0', c['out'])
            if 'dim_size' in c:
                store(base + '_dim_size', c['dim_size'])
            if op in ('min', 'max'):
                store(key + '_res', res[0])
                store(key + '_arg', res[1])
            else:
                store(key + '_res', res)
            META.append(key)
        # gather_coo of the reduced result back to the sources
        red = P.segment_sum_coo(c['src'], c['index'], None, c.get('dim_size'))
        gat = None
        if c['index'].dim() == 1 or c['index'].size(0) == red.size(0):
            gat = P.gather_coo(red, c['index'], None)
        if gat is not None:
            key = f'gather{ci}'
            store(key + '_src', red)
            store(key + '_index', c['index'])
            store(key + '_res', gat)
            META.append(key)


def sort_cases():
    g = torch.Generator().manual_seed(2)
    cases = [
        torch.randperm(40_000, generator=g),                                   # test_index_sort.py:28-33 (radix path: n > 32768)
        torch.randint(0, 50, (40_000,), generator=g),                           # heavy duplicates (stability)
        torch.randint(0, 2**31 - 1, (33_000,), generator=g).to(torch.int32),
        torch.randint(0, 2**40, (33_000,), generator=g),
        torch.randint(0, 200, (40_000,), generator=g).to(torch.int16),
        torch.randint(0, 200, (40_000,), generator=g).to(torch.uint8),
        torch.randint(0, 1000, (1000,), generator=g),                           # below GRAIN_SIZE: at::sort path
        torch.zeros(0, dtype=torch.long),
        torch.tensor([7]),
    ]
    for ci, keys in enumerate(cases):
        for with_max in (False, True):
            mx = None
            if with_max:
                if keys.numel() == 0:
                    continue
                mx = int(keys.max()) * 3 + 5  # an over-estimate is allowed
            vals, idx = P.index_sort(keys, mx)
            if keys.numel() <= 32768:
                # at::sort is not guaranteed stable; the documented contract is torch.sort(stable=True)
                vals, idx = torch.sort(keys, stable=True)
            key = f'sort{ci}_{int(with_max)}'
            store(f'sort{ci}_keys', keys)
            if mx is not None:
                store(key + '_max', mx)
            if f'sort{ci}_idx' in D:
                assert (D[f'sort{ci}_idx'] == idx.numpy()).all()  # `max` only sets the pass count
            store(f'sort{ci}_idx', idx)  # vals == keys[idx]
            META.append(key)


if __name__ == '__main__':
    scatter_cases()
    coo_cases()
    sort_cases()
    D['__cases__'] = np.array(META)
    np.savez_compressed(osp.join(HERE, 'reduce_golden.npz'), **D)
    print(f'{len(META)} cases ->', osp.join(HERE, 'reduce_golden.npz'))
