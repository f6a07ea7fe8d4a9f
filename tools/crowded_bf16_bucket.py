"""5000 bf16 values of -2 ... 2 per slice into ONE bucket through the packed 16-bit atomics (what tools/fuzz_reduce.py seed 7 case 174
does): how far from the exact sum, how often, hardware atomics and the compare-and-swap flavour; next to the largest partial sum of
the sequential order (bf16 holds integers exactly up to 256).   python tools/crowded_bf16_bucket.py [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyg_lib_amd import _capi, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
for mode in (0, 1):
    _capi.lib().pyg_hip_set_float_atomic_mode(mode)
    hist = {}
    worst_partial = 0
    for r in range(reps):
        vals = torch.randint(-2, 3, (3, 5000, 6), generator=g).double()
        want = vals.sum(1, keepdim=True)
        if want.abs().max() > 256:
            continue
        worst_partial = max(worst_partial, float(vals.cumsum(1).abs().max()))
        idx = torch.zeros(3, 5000, 6, dtype=torch.long)
        got = ops.scatter_sum(vals.bfloat16().to(dev), idx.to(dev), 1, None, 1)
        d = float((got.double().cpu() - want).abs().max())
        hist[d] = hist.get(d, 0) + 1
    print('float atomics =', 'cas' if mode else 'hw', '| max |diff| histogram:', dict(sorted(hist.items())), '| largest sequential partial sum', worst_partial)
_capi.lib().pyg_hip_set_float_atomic_mode(0)
