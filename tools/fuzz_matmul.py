"""Random-shape fuzz of segment_matmul / grouped_matmul against float64 products: every dispatch family (ticket, item
ring, register-W, split-bf16, fp32 MFMA, LDS-W, general shapes, one-thread-per-output), ragged / empty / one-row
segments, bias, transposed weights, device and host `ptr`.   python tools/fuzz_matmul.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyg_lib_amd import ops

dev = 'cuda:0'
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
g = torch.Generator(device=dev).manual_seed(seed)
TOL = {torch.bfloat16: 2 ** -6, torch.float16: 2 ** -8, torch.float32: 1e-5}
seen = {}
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    dtype = [torch.bfloat16, torch.float16, torch.float32][int(rng.integers(0, 3))]
    K = int(rng.choice([32, 64, 128, 128, 128, 256, 256, 512, 100, 40, 129]))
    M = int(rng.choice([32, 64, 128, 128, 256, 256, 384, 96, 100]))
    B = int(rng.choice([1, 2, 7, 33, 150, 600]))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        sizes = rng.integers(0, 40, B)
    elif kind == 1:
        sizes = rng.integers(0, 300, B)
    elif kind == 2:
        sizes = rng.integers(0, 5000, B)
    elif kind == 3:
        sizes = (rng.random(B) < 0.4) * rng.integers(1, 3000, B)
    else:
        sizes = np.array([int(rng.integers(1, 300_000))] + [int(v) for v in rng.integers(0, 70, B - 1)])
    if sizes.sum() > 1_500_000:
        sizes = sizes // 4
    ptr = torch.tensor([0] + np.cumsum(sizes).tolist())
    n = int(ptr[-1])
    x = torch.randn(n, K, device=dev, generator=g).to(dtype)
    w = (torch.randn(B, K, M, device=dev, generator=g) / K ** 0.5).to(dtype)
    trans = bool(rng.integers(0, 2))
    if trans:
        w = w.transpose(1, 2).contiguous().transpose(1, 2)
    bias = torch.randn(B, M, device=dev, generator=g).to(dtype) if rng.integers(0, 2) else None
    sched = str(rng.choice(['auto', 'auto', 'auto', 'contiguous', 'ticket', 'ring', 'cyclic']))
    split = bool(rng.integers(0, 4))
    grouped = bool(rng.integers(0, 3) == 0) and B <= 150
    try:
        ops.set_matmul_schedule(sched)
        torch.set_float32_matmul_precision('high' if split else 'highest')
        if grouped:
            xs = [x[int(ptr[b]):int(ptr[b + 1])] for b in range(B)]
            ws = [w[b] for b in range(B)]
            outs = ops.grouped_matmul(xs, ws, None if bias is None else [bias[b] for b in range(B)])
            out = torch.cat(outs) if n else x.new_zeros(0, M)
        else:
            p = ptr.to(dev) if rng.integers(0, 2) else ptr
            out = ops.segment_matmul(x, p, w, bias)
        var = ops.matmul_last_variant()
    finally:
        ops.set_matmul_schedule('auto')
        torch.set_float32_matmul_precision('highest')
    seen[var] = seen.get(var, 0) + 1
    assert out.shape == (n, M) and out.dtype == dtype
    if n == 0:
        continue
    # float64 reference, one relation at a time (bounded memory)
    worst = 0.0
    wd = w.double()
    for b in range(B):
        s0, e0 = int(ptr[b]), int(ptr[b + 1])
        if e0 == s0:
            continue
        ref = x[s0:e0].double() @ wd[b]
        if bias is not None:
            ref = ref + bias[b].double()
        err = (out[s0:e0].double() - ref).abs().max().item()
        worst = max(worst, err / (ref.abs().max().item() + 1e-9))
    assert worst <= 4 * TOL[dtype], (it, dtype, K, M, B, kind, trans, bias is not None, sched, split, grouped, var, worst)
print('fuzz_matmul: %d cases, variants:' % it, dict(sorted(seen.items())))
