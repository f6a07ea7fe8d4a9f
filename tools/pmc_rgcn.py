"""The fused R-GCN layer of one C5 batch, a few times (for rocprofv3 --pmc passes):  python tools/pmc_rgcn.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.bench_rgcn as B
from pyg_lib_amd import rgcn, sampler
dev = torch.device('cuda', 0)
types = list(B.SIZES)
ets = [(s, r, d) for s, r, d, _ in B.RELS]
rp, cl = B.make_graph(dev)
feat = {t: torch.randn(B.SIZES[t], 128, device=dev).bfloat16() for t in types}
W = (torch.randn(len(ets), 128, 128, device=dev) / 128 ** 0.5).bfloat16()
torch.manual_seed(1)
seeds = torch.randperm(B.SIZES['paper'])[:1024].to(dev)
out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds}, {e: [15, 10] for e in ets})
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    y = rgcn.rgcn_layer_fused_tables(feat, out[2], types, out[0], out[1], ets, W, grouped=False)   # the atomic kernel
    y = rgcn.rgcn_layer_fused_tables(feat, out[2], types, out[0], out[1], ets, W, grouped=True)    # row-start + atomic-free kernel
torch.cuda.synchronize()
