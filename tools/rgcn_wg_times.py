"""Per-workgroup timeline of the atomic-free layer kernel on the C5 sample (a library built with -DPYG_HIP_RGCN_ABLATE=16 writes
start / end stamps, sub-items and iterations of every workgroup into the last 64 KB of the workspace):
    python tools/rgcn_wg_times.py [F]"""
import ctypes
import sys

import torch

import bench_legs
from pyg_lib_amd import sampler, rgcn, _capi

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda:0')
types = list(bench_legs.MAG_SIZES)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
rp, cl = bench_legs.make_mag_graph(dev)
feat = {t: torch.randn(bench_legs.MAG_SIZES[t], F, device=dev).bfloat16() for t in types}
W = (torch.randn(len(ets), F, F, device=dev) / F ** 0.5).bfloat16()
seeds = torch.randperm(bench_legs.MAG_SIZES['paper'], generator=torch.Generator().manual_seed(1))[:1024].to(dev)
torch.manual_seed(100)
out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds}, {e: [15, 10] for e in ets})
row_d, col_d, node_d = out[0], out[1], out[2]
off = rgcn.type_offsets({t: node_d[t].numel() for t in types}, types)
n = off['__total__']


class Rel(ctypes.Structure):
    _fields_ = [('gather_index', ctypes.c_void_p), ('scatter_index', ctypes.c_void_p), ('num_edges', ctypes.c_int64),
                ('gather_offset', ctypes.c_int64), ('scatter_offset', ctypes.c_int64), ('weight', ctypes.c_void_p),
                ('x', ctypes.c_void_p), ('gather_map', ctypes.c_void_p), ('x_rows', ctypes.c_int64), ('gather_map_len', ctypes.c_int64),
                ('scatter_rows', ctypes.c_int64)]


L = _capi.lib()
L.pyg_hip_rgcn_grouped_workspace_size.restype = ctypes.c_size_t
L.pyg_hip_rgcn_grouped_workspace_size.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
L.pyg_hip_rgcn_fused.restype = ctypes.c_int
L.pyg_hip_rgcn_fused.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                 ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                 ctypes.c_void_p]
rels = (Rel * len(ets))()
for i, (s, r, d) in enumerate(ets):   # csc=False: gather col (dst type), scatter row (src type)
    g, sc = col_d[(s, r, d)], row_d[(s, r, d)]
    rels[i] = Rel(g.data_ptr(), sc.data_ptr(), g.numel(), 0, off[s], W[i].data_ptr(), feat[d].data_ptr(), node_d[d].data_ptr(),
                  feat[d].size(0), node_d[d].numel(), 0)
need = L.pyg_hip_rgcn_grouped_workspace_size(ctypes.addressof(rels), len(ets), n)
ws = torch.zeros(need, dtype=torch.uint8, device=dev)
y = torch.empty(n, F, dtype=torch.bfloat16, device=dev)
for _ in range(5):
    rc = L.pyg_hip_rgcn_fused(3, None, 0, ctypes.addressof(rels), len(ets), y.data_ptr(), n, F, F, 8, ws.data_ptr(), need,
                              torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.pyg_hip_last_error()
torch.cuda.synchronize()
d = ws[-65536:].view(torch.int64).cpu().view(-1, 4)
G = int((d[:, 1] > 0).sum())
d = d[:G]
t0 = int(d[:, 0].min())
start, end = (d[:, 0] - t0).float() / 100.0, (d[:, 1] - t0).float() / 100.0     # us (100 MHz)
items, iters = d[:, 2], d[:, 3]
print(f'F = {F}: {G} workgroups; kernel span {end.max():.1f} us; starts {start.min():.1f} ... {start.max():.1f}; '
      f'ends: min {end.min():.1f} median {end.median():.1f} max {end.max():.1f}')
print('sub-items per workgroup: mean %.2f max %d min %d; iterations mean %.2f max %d' %
      (items.float().mean(), items.max(), items.min(), iters.float().mean(), iters.max()))
for k in sorted(set(items.tolist())):
    m = items == k
    dur = (end - start)[m]
    print(f'  {k:3d} sub-items: {int(m.sum()):4d} workgroups, duration mean {dur.mean():6.1f} us (min {dur.min():.1f} max {dur.max():.1f}), '
          f'{dur.mean() / max(k + 3, 1):.2f} us per iteration')
