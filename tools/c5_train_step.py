"""Forward + backward of the fused R-GCN layer on one C5 sample (for rocprofv3 --kernel-trace --stats):
    python tools/c5_train_step.py [iters] [atomic]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
from pyg_lib_amd import sampler, rgcn
dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rgcn.set_dx_mode('atomic' if len(sys.argv) > 2 and sys.argv[2] == 'atomic' else 'grouped')
types = list(bench_legs.MAG_SIZES)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
rp, cl = bench_legs.make_mag_graph(dev)
F = 128
feat = {t: torch.randn(bench_legs.MAG_SIZES[t], F, device=dev).bfloat16() for t in types}
W = (torch.randn(len(ets), F, F, device=dev) / F ** 0.5).bfloat16()
seeds = torch.randperm(bench_legs.MAG_SIZES['paper'])[:1024].to(dev)
out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds}, {e: [15, 10] for e in ets})
off = rgcn.type_offsets({t: out[2][t].numel() for t in types}, types)
x = torch.cat([feat[t][out[2][t]] for t in types])
c = torch.randn(off['__total__'], F, device=dev).bfloat16()


def step():
    xg, wg = x.detach().requires_grad_(), W.detach().requires_grad_()
    (rgcn.rgcn_layer_fused(xg, off, out[0], out[1], ets, wg) * c).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
print('ms per forward+backward: %.4f (dx %s)' % ((time.perf_counter() - t0) / iters * 1e3, rgcn._DX_MODE[0]))
