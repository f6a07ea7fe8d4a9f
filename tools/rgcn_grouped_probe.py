"""C5 layer alone, both kernels (atomic adds into a zero-filled output / atomic-free grouped), for rocprofv3:
python tools/rgcn_grouped_probe.py [iters] [fan-out, e.g. 25,10] [F: 128 | 256] [f32]"""
import sys
import time

import torch

import bench_legs
from pyg_lib_amd import sampler, rgcn

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fan = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [15, 10]
dev = torch.device('cuda:0')
types = list(bench_legs.MAG_SIZES)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
rp, cl = bench_legs.make_mag_graph(dev)
F = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dt = torch.float32 if len(sys.argv) > 4 and sys.argv[4] == 'f32' else torch.bfloat16
feat = {t: torch.randn(bench_legs.MAG_SIZES[t], F, device=dev).to(dt) for t in types}
W = (torch.randn(len(ets), F, F, device=dev) / F ** 0.5).to(dt)
seeds = torch.randperm(bench_legs.MAG_SIZES['paper'], generator=torch.Generator().manual_seed(1))[:1024].to(dev)
torch.manual_seed(100)
out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds}, {e: fan for e in ets})
print('fan-out', fan, 'edges', sum(v.numel() for v in out[0].values()), 'nodes', sum(v.numel() for v in out[2].values()))
for grouped in (False, True):
    f = lambda: rgcn.rgcn_layer_fused_tables(feat, out[2], types, out[0], out[1], ets, W, grouped=grouped)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    print(f'grouped={grouped}: {(time.perf_counter() - t0) / iters * 1e3:.4f} ms per layer', flush=True)
if F != 128 or dt == torch.float32:   # (grouped=False above was the three-op chain: the atomic kernel is 16-bit, 128 x 128)
    print('(grouped=False = the three-op chain for this width)')
e = sum(v.numel() for v in out[0].values())
n = sum(v.numel() for v in out[2].values())
esz = 4 if dt == torch.float32 else 2
print('formula bytes', e * (F * esz + 16) + n * F * esz + len(ets) * F * F * esz)
print('pending', rgcn.pending_index_error())
