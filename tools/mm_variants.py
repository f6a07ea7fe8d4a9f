"""Within-process interleaved A/B of segment_matmul kernel variants on C2 (experiment harness)."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyg_lib_amd import ops, _capi

dev = torch.device('cuda:0')
x, ptr, w, (N, B, F) = bench.make_c2(dev, 0, 1)
L = _capi.lib()
L.pyg_hip_profile_enable.argtypes = [ctypes.c_int]
L.pyg_hip_profile_collect.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.pyg_hip_profile_collect.restype = ctypes.c_int
alg = 2 * (2 * N * F) + 2 * B * F * F + 8 * (B + 1)
variants = [dict(PYG_HIP_MM_FLAGS='3', PYG_HIP_MM_WGS='2', PYG_HIP_MM_CHUNK=str(c)) for c in (0, 1, 2, 4, 8, 16, 64)]
variants += [dict(PYG_HIP_MM_FLAGS='3', PYG_HIP_MM_WGS='8', PYG_HIP_MM_CHUNK=str(c)) for c in (1, 4)]
res = {i: [] for i in range(len(variants))}
for rnd in range(6):
    for i, v in enumerate(variants):
        os.environ.update(v)
        ops.segment_matmul(x, ptr, w)
        torch.cuda.synchronize()
        L.pyg_hip_profile_enable(1)
        for _ in range(5):
            ops.segment_matmul(x, ptr, w)
        buf = (ctypes.c_float * 8)()
        n = L.pyg_hip_profile_collect(buf, 8)
        L.pyg_hip_profile_enable(0)
        res[i] += [buf[j] for j in range(n)]
for i, v in enumerate(variants):
    a = np.array(res[i])
    print(v, f'median {np.median(a):.4f} ms min {a.min():.4f} ms -> {alg / np.median(a) / 1e6:.1f} GB/s (best {alg / a.min() / 1e6:.1f})')

# reference points on the same box: plain device copy (read+write) and fill (write only)
y = torch.empty_like(x)
def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n
ms = timeit(lambda: y.copy_(x))
print(f'torch copy_ {x.numel() * 2 * 2 / ms / 1e6:.1f} GB/s ({ms:.3f} ms)')
ms = timeit(lambda: y.zero_())
print(f'torch zero_ {x.numel() * 2 / ms / 1e6:.1f} GB/s ({ms:.3f} ms)')
ms = timeit(lambda: x.sum())
print(f'torch sum (read only) {x.numel() * 2 / ms / 1e6:.1f} GB/s ({ms:.3f} ms)')
xf = x.view(torch.float32)
ms = timeit(lambda: torch.add(xf, 1.0, out=y.view(torch.float32)))
print(f'torch add fp32 view {x.numel() * 2 * 2 / ms / 1e6:.1f} GB/s ({ms:.3f} ms)')
