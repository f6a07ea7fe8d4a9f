"""Differential fuzz of segment_{sum,mean,min,max}_csr / gather_csr / softmax_csr against the oracle on integer-valued data (sums
exact in every dtype and every order): random row-length distributions WITH hub rows (the row / LDS-streamed / hub-chunk / long
kernels), row widths 1 ... 300, leading (batched) dims with shared or per-slice offsets, fresh or given `out`.
python tools/fuzz_csr.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from pyg_lib_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = 'cuda:0'
OPS = {'sum': oracle.CSR_SUM, 'mean': oracle.CSR_MEAN, 'min': oracle.CSR_MIN, 'max': oracle.CSR_MAX}
bad = 0
for case in range(cases):
    dtype = [torch.float32, torch.bfloat16, torch.float16, torch.float64, torch.int32, torch.int64][rng.integers(0, 6)]
    K = int([1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 64, 100, 128, 300][rng.integers(0, 14)])
    rows = int([1, 2, 7, 100, 3000, 20_000][rng.integers(0, 6)])
    mean = [0.5, 2, 8, 20, 70, 400][rng.integers(0, 6)]
    lens = rng.poisson(mean, rows)
    for _ in range(int(rng.integers(0, 4))):   # hubs
        lens[rng.integers(0, rows)] = int([600, 3000, 5000, 30_000][rng.integers(0, 4)])
    while lens.sum() * K > 6_000_000:
        lens = lens // 2
    lead = int([1, 1, 1, 2, 3][rng.integers(0, 5)])
    E = int(lens.sum())
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    per_slice = lead > 1 and rng.integers(0, 2) == 1
    if per_slice:   # every slice its own offsets (same total)
        ips = [indptr]
        for _ in range(lead - 1):
            cut = np.sort(rng.integers(0, E + 1, rows - 1)) if rows > 1 else np.zeros(0, np.int64)
            ips.append(np.concatenate([[0], cut, [E]]).astype(np.int64))
        indptr_np = np.stack(ips)
    else:
        indptr_np = indptr[None, :] if lead > 1 else indptr   # shared offsets: broadcast over the leading dim
    shape = ([lead] if lead > 1 else []) + [E, K]
    src = torch.from_numpy(rng.integers(-6, 7, shape).astype(np.float32)).to(dtype)
    bf16 = dtype == torch.bfloat16
    src_np = src.view(torch.int16).numpy().view(np.uint16) if bf16 else src.numpy()
    code = oracle.BF16 if bf16 else None
    ip = torch.from_numpy(indptr_np).to(dev)
    tag = f'case {case}: {dtype} K={K} rows={rows} E={E} lead={lead} per_slice={per_slice} max_len={int(lens.max()) if rows else 0}'
    try:
        for op in ('sum', 'mean', 'min', 'max'):
            if op == 'mean' and not dtype.is_floating_point:
                continue
            given = rng.integers(0, 3) == 0 and op != 'mean'
            out_shape = ([lead] if lead > 1 else []) + [rows, K]
            base = torch.from_numpy(rng.integers(-3, 4, out_shape).astype(np.float32)).to(dtype) if given else None
            base_np = None if base is None else (base.view(torch.int16).numpy().view(np.uint16) if bf16 else base.numpy())
            want, warg = oracle.segment_csr(OPS[op], src_np, indptr_np, base_np, code)
            res = getattr(ops, f'segment_{op}_csr')(src.to(dev), ip, None if base is None else base.clone().to(dev))
            val = (res[0] if op in ('min', 'max') else res).cpu()
            want_t = torch.from_numpy(want)
            if bf16:
                want_t = want_t.view(torch.int16).view(torch.bfloat16)
            if op == 'mean':
                ok = torch.allclose(val.double(), want_t.double(), rtol=2 ** -7 if dtype in (torch.bfloat16, torch.float16) else 1e-6, atol=1e-6)
            else:
                ok = torch.equal(val.double(), want_t.double())
            if op in ('min', 'max'):
                ok = ok and torch.equal(res[1].cpu(), torch.from_numpy(warg))
            if not ok:
                bad += 1
                print('MISMATCH', op, 'given out' if given else 'fresh', tag, flush=True)
        rows_t = torch.from_numpy(rng.integers(-50, 50, ([lead] if lead > 1 else []) + [rows, K]).astype(np.float32)).to(dtype)
        got = ops.gather_csr(rows_t.to(dev), ip).cpu()
        rows_np = rows_t.view(torch.int16).numpy().view(np.uint16) if bf16 else rows_t.numpy()
        zero = np.zeros(shape, dtype=rows_np.dtype)
        want = torch.from_numpy(oracle.gather_csr(rows_np, indptr_np, zero, code))
        if bf16:
            want = want.view(torch.int16).view(torch.bfloat16)
        if not torch.equal(got.double(), want.double()):
            bad += 1
            print('MISMATCH gather', tag, flush=True)
        if dtype == torch.float32 and not per_slice and E > 0:
            x = (rng.standard_normal(shape) * 3).astype(np.float32)
            dim = 1 if lead > 1 else 0
            y = ops.softmax_csr(torch.from_numpy(x).to(dev), torch.from_numpy(indptr).to(dev), dim).cpu()   # (ptr is 1-D there)
            w = torch.from_numpy(oracle.softmax_csr(x, indptr, dim))
            if not torch.allclose(y, w, rtol=3e-4, atol=1e-9):
                bad += 1
                print('MISMATCH softmax', tag, float((y - w).abs().max()), flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print('ERROR', tag, repr(e)[:300], flush=True)
torch.cuda.synchronize()
print(f'fuzz_csr: {cases} cases, seed {seed}: {bad} bad')
