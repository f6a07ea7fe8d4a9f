"""segment_{sum,min,max}_csr / gather_csr on int8 / uint8 / int16 rows with hub rows against the oracle (sums of 0 / 1 values wrap like the reference).  python tools/small_int_csr_check.py"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from pyg_lib_amd import ops
rng = np.random.default_rng(5)
OPS = {'sum': oracle.CSR_SUM, 'min': oracle.CSR_MIN, 'max': oracle.CSR_MAX}
bad = 0
for dtype in (torch.int8, torch.uint8, torch.int16):
    for K in (1, 3, 16, 32, 48):
        lens = rng.integers(0, 9, 3000); lens[[7, 900]] = [700, 5000]
        ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        E = int(ip[-1])
        src = torch.from_numpy(rng.integers(0, 2, (E, K))).to(dtype)
        for op in ('sum', 'min', 'max'):
            want, warg = oracle.segment_csr(OPS[op], src.numpy(), ip)
            res = getattr(ops, f'segment_{op}_csr')(src.cuda(), torch.from_numpy(ip).cuda())
            val = res if op == 'sum' else res[0]
            ok = np.array_equal(val.cpu().numpy(), want) and (op == 'sum' or np.array_equal(res[1].cpu().numpy(), warg))
            if not ok:
                bad += 1; print('MISMATCH', dtype, K, op)
        rows = torch.from_numpy(rng.integers(0, 100, (3000, K))).to(dtype)
        g = ops.gather_csr(rows.cuda(), torch.from_numpy(ip).cuda()).cpu()
        if not torch.equal(g, torch.repeat_interleave(rows, torch.from_numpy(lens), dim=0)):
            bad += 1; print('MISMATCH gather', dtype, K)
print('small ints:', bad, 'bad')
