"""C5 hetero sampler: wall time per call of (a) the Python front, (b) the operator with prebuilt Dict arguments;
PYG_HIP_SAMPLER_TRACE=1 (set it in the environment) adds the library's own host timeline."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
from pyg_lib_amd import sampler
dev = torch.device('cuda:0')
rp, cl = bench_legs.make_mag_graph(dev)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
fan = {e: [15, 10] for e in ets}
g = torch.Generator().manual_seed(1)
seeds = [torch.randperm(bench_legs.MAG_SIZES['paper'], generator=g)[:1024].to(dev) for _ in range(40)]
nt = sorted({t for e in ets for t in (e[0], e[-1])})
rel = sampler._to_rel_keys


def front(s):
    return sampler.hetero_neighbor_sample(rp, cl, {'paper': s}, fan)


def dict_op(s):
    return torch.ops.pyg.hetero_neighbor_sample(nt, ets, rel(rp), rel(cl), {'paper': s}, rel(fan), None, None, None, None, False,
                                                False, True, False, 'uniform', True)


for name, fn in (('front', front), ('dict op', dict_op), ('front', front), ('dict op', dict_op)):
    for s in seeds[:8]:
        fn(s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for s in seeds[8:]:
        fn(s)
    torch.cuda.synchronize()
    print('%-8s %.1f us per call' % (name, (time.perf_counter() - t) / 32 * 1e6), flush=True)
