"""Experiment builds only (EXTRA_HIPCC_FLAGS=-DPYG_HIP_FOLD_TIMING): phases of the seeds launch of a C3 batch, 100 MHz stamps."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler
from pyg_lib_amd import sampler, _capi
dev = torch.device('cuda:0')
rowptr, col = bench_sampler.make_graph(dev)
g = torch.Generator().manual_seed(1)
seeds = torch.randperm(bench_sampler.N_NODES, generator=g)[:1024 * 8].to(dev).view(8, 1024)
L = _capi.lib()
buf = (ctypes.c_ulonglong * 16)()
kb = (ctypes.c_ulonglong * 128)()
L.pyg_hip_debug_kstamps(kb)
for b in range(8):
    sampler.neighbor_sample(rowptr, col, seeds[b], [15, 10, 5])
    torch.cuda.synchronize()
    L.pyg_hip_debug_fold_stamps(buf)
    t = [buf[i] for i in range(8)]
    print('batch', b, 'us since entry:', ' '.join('%.2f' % ((x - t[0]) / 100.0) for x in t))
    n = L.pyg_hip_debug_kstamps(kb)
    ev = sorted((kb[2 * i + 1], kb[2 * i]) for i in range(min(n, 64)))
    print('   kernel entries (id: 1xx sample GMAX, 2NM scan MAXNC MODE, 999 = seeds kernel end):',
          ' '.join('%d@%.1f' % (i, (t - ev[0][0]) / 100.0) for t, i in ev))
