"""Where the fused R-GCN layer's time goes on one C5 batch (experiment build: EXTRA_HIPCC_FLAGS=-DPYG_HIP_EXPERIMENTS):
the layer alone under PYG_HIP_RGCN_WGS (workgroups per CU of the persistent grid), PYG_HIP_RGCN_MIN_TILES (tiles per
workgroup at least) and PYG_HIP_RGCN_DBG (1 = no atomics), the zero fill of the output on its own, and the kernel
without the fill."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.bench_rgcn as B  # noqa: E402
from pyg_lib_amd import rgcn, sampler  # noqa: E402

dev = torch.device('cuda', 0)
types = list(B.SIZES)
ets = [(s, r, d) for s, r, d, _ in B.RELS]
rp, cl = B.make_graph(dev)
feat = {t: torch.randn(B.SIZES[t], 128, device=dev).bfloat16() for t in types}
W = (torch.randn(len(ets), 128, 128, device=dev) / 128 ** 0.5).bfloat16()
torch.manual_seed(1)
seeds = torch.randperm(B.SIZES['paper'])[:1024].to(dev)
out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds}, {e: [15, 10] for e in ets})
edges = sum(v.numel() for v in out[0].values())
nodes = sum(v.numel() for v in out[2].values())
print('edges', edges, 'nodes', nodes, flush=True)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print('zero fill of the output alone: %.1f us' % timed(lambda: feat['paper'].new_zeros(nodes, 128)))
pre = feat['paper'].new_zeros(nodes, 128)
tidx = {t: i for i, t in enumerate(types)}
off = rgcn.type_offsets({t: out[2][t].numel() for t in types}, types)
args = ([feat[t] for t in types], [out[2][t] for t in types], [tidx[e[2]] for e in ets], [out[1][e] for e in ets],
        [out[0][e] for e in ets], [off[e[0]] for e in ets], W)
for wgs in ('2',):
    for mt in ('2',):
        for dbg in ('0', '1', '2', '8'):  # 1 no atomics, 2 no row gathers, 8 no scatter walk
            os.environ['PYG_HIP_RGCN_WGS'] = wgs
            os.environ['PYG_HIP_RGCN_MIN_TILES'] = mt
            os.environ['PYG_HIP_RGCN_DBG'] = dbg
            t_layer = timed(lambda: rgcn.rgcn_layer_fused_tables(feat, out[2], types, out[0], out[1], ets, W, grouped=False))
            t_kernel = timed(lambda: torch.ops.pyg.rgcn_fused_tables(*args, pre))
            print(f'wgs/CU {wgs} min tiles {mt} dbg {dbg}: layer {t_layer:7.1f} us   op without the fill {t_kernel:7.1f} us', flush=True)
