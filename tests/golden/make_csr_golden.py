"""Generates tests/golden/csr_golden.npz with the REAL reference (build container only).

    bash oracle/build_ref.sh && python tests/golden/make_csr_golden.py

Loads oracle/_ref/libpyg_ref.so -- the reference's own segment_*_csr / gather_csr / softmax_csr CPU
kernels compiled unmodified from /root/reference by oracle/build_ref.sh -- and records what
`torch.ops.pyg.*` returns on CPU tensors for a battery of inputs shaped after the reference's
test/ops/test_segment_csr.py and test/ops/test_softmax.py (dtypes, 1-D, K=1, large K, broadcast
indptr, empty rows, all-empty, huge + short rows, out=, ties).  Inputs and outputs are stored
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


def csr_cases():
    g = torch.Generator().manual_seed(3)
    ptr6 = torch.tensor([0, 2, 5, 5, 6])  # 4 rows over 6 sources, one empty row in the middle
    cases = []
    for dt in (torch.float32, torch.float64, torch.int64, torch.int32, torch.bfloat16, torch.float16):
        cases.append(dict(src=rand((6, 4), dt, g), indptr=ptr6))
    cases.append(dict(src=rand((6,), torch.float32, g), indptr=ptr6))                       # 1-D, K = 1
    cases.append(dict(src=rand((6, 1), torch.float32, g), indptr=ptr6))                     # trailing 1
    cases.append(dict(src=rand((6, 3, 5), torch.float32, g), indptr=ptr6))                  # K = 15
    cases.append(dict(src=rand((6, 128), torch.float32, g), indptr=ptr6))                   # large K
    cases.append(dict(src=rand((6, 4), torch.float32, g), indptr=torch.tensor([0, 0, 0, 0])))  # all rows empty
    cases.append(dict(src=rand((6, 4), torch.float32, g), indptr=torch.tensor([1, 3, 3, 5])))  # uncovered ends
    # 2-D indptr: one CSR per leading slice, and a [1, R+1] indptr broadcast over the slices
    cases.append(dict(src=rand((3, 6, 2), torch.float32, g),
                      indptr=torch.tensor([[0, 2, 5, 6], [0, 0, 3, 6], [0, 6, 6, 6]])))
    cases.append(dict(src=rand((3, 6, 2), torch.float32, g), indptr=torch.tensor([[0, 2, 5, 6]])))
    # out= (sum accumulates into it; min/max continue from it; a tie with out keeps the sentinel arg)
    cases.append(dict(src=rand((6, 4), torch.float32, g), indptr=ptr6, out=rand((4, 4), torch.float32, g)))
    cases.append(dict(src=torch.tensor([2., 3., 2., 7., -1., 2.]), indptr=ptr6,
                      out=torch.tensor([2., 9., -5., 2.])))
    # ties (first match) and negative values
    cases.append(dict(src=torch.tensor([[1., 1.], [1., -2.], [3., -2.], [3., 5.], [1., 5.], [0., 0.]]), indptr=ptr6))
    cases.append(dict(src=torch.tensor([5, 5, 7, 7, 7, 1], dtype=torch.int64), indptr=ptr6))
    # stress: many short rows + a few huge ones (test_segment_csr.py:261-281)
    lens = torch.cat([torch.randint(0, 5, (300,), generator=g), torch.tensor([4000, 0, 2500]),
                      torch.randint(0, 5, (200,), generator=g)])
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    cases.append(dict(src=rand((int(ptr[-1]), 8), torch.float32, g), indptr=ptr))
    cases.append(dict(src=rand((int(ptr[-1]),), torch.float32, g), indptr=ptr))
    cases.append(dict(src=rand((int(ptr[-1]), 8), torch.bfloat16, g), indptr=ptr))
    cases.append(dict(src=rand((int(ptr[-1]), 3), torch.int32, g), indptr=ptr))
    cases.append(dict(src=rand((0, 4), torch.float32, g), indptr=torch.tensor([0, 0, 0])))   # empty input
    for ci, c in enumerate(cases):
        src, indptr = c['src'], c['indptr']
        base = f'csr{ci}'
        store(base + '_src', src)
        store(base + '_indptr', indptr)
        if 'out' in c:
            store(base + '_out0', c['out'])
        for op in ('sum', 'mean', 'min', 'max'):
            if op == 'mean' and not src.is_floating_point():
                continue
            out = c['out'].clone() if 'out' in c else None
            res = getattr(P, f'segment_{op}_csr')(src, indptr, out)
            key = f'{base}_{op}'
            if op in ('min', 'max'):
                store(key + '_res', res[0])
                store(key + '_arg', res[1])
            else:
                store(key + '_res', res)
            META.append(key)
        # gather_csr of the row sums back to the source positions (the backward of segment_sum_csr);
        # positions outside [indptr[0], indptr[-1]) are left untouched, so a defined out= is passed
        if src.numel() > 0:
            red = P.segment_sum_csr(src, indptr, None)
            buf = torch.full_like(src, 77)
            gat = P.gather_csr(red, indptr, buf)
            key = f'{base}_gather'
            store(key + '_src', red)
            store(key + '_out0', torch.full_like(src, 77))
            store(key + '_res', gat)
            META.append(key)


def softmax_cases():
    g = torch.Generator().manual_seed(4)
    cases = [
        dict(src=rand((8, 3), torch.float32, g), ptr=torch.tensor([0, 3, 4, 7, 8]), dim=0),   # test_softmax.py shape
        dict(src=rand((8,), torch.float32, g), ptr=torch.tensor([0, 3, 4, 7, 8]), dim=0),
        dict(src=rand((2, 8, 3), torch.float32, g), ptr=torch.tensor([0, 5, 5, 8]), dim=1),   # empty group, outer > 1
        dict(src=rand((4, 6), torch.float32, g), ptr=torch.tensor([0, 2, 6]), dim=1),         # last dim
        dict(src=rand((500, 4), torch.float32, g) * 20,
             ptr=torch.cat([torch.zeros(1, dtype=torch.long),
                            torch.sort(torch.randint(0, 500, (60,), generator=g)).values,
                            torch.tensor([500])]), dim=0),
    ]
    for ci, c in enumerate(cases):
        src, ptr, dim = c['src'], c['ptr'], c['dim']
        out = P.softmax_csr(src, ptr, dim)
        og = rand(tuple(src.shape), torch.float32, g)
        gin = P.softmax_csr_backward(out, og, ptr, dim)
        key = f'softmax{ci}'
        store(key + '_src', src)
        store(key + '_ptr', ptr)
        store(key + '_dim', dim)
        store(key + '_res', out)
        store(key + '_out_grad', og)
        store(key + '_in_grad', gin)
        META.append(key)


if __name__ == '__main__':
    csr_cases()
    softmax_cases()
    D['__cases__'] = np.array(META)
    np.savez_compressed(osp.join(HERE, 'csr_golden.npz'), **D)
    print(f'{len(META)} cases ->', osp.join(HERE, 'csr_golden.npz'))
