"""Items (16-row block, relation with edges into it) per workgroup of the atomic-free layer's persistent launch on the C5
sample: static round-robin (block b -> workgroup b % G) against the mean.  python tools/rgcn_item_balance.py [G]"""
import sys

import torch

import bench_legs
from pyg_lib_amd import sampler, rgcn

G = int(sys.argv[1]) if len(sys.argv) > 1 else 768
dev = torch.device('cuda:0')
types = list(bench_legs.MAG_SIZES)
ets = [(s, r, d) for s, r, d, _ in bench_legs.MAG_RELS]
rp, cl = bench_legs.make_mag_graph(dev)
seeds = torch.randperm(bench_legs.MAG_SIZES['paper'], generator=torch.Generator().manual_seed(1))[:1024].to(dev)
torch.manual_seed(100)
out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds}, {e: [15, 10] for e in ets})
row_d, node_d = out[0], out[2]
off = rgcn.type_offsets({t: node_d[t].numel() for t in types}, types)
total = off['__total__']
nblocks = (total + 15) // 16
items = torch.zeros(nblocks, dtype=torch.long, device=dev)
for (s, r, d) in ets:
    blk = torch.unique((row_d[(s, r, d)] + off[s]) // 16)
    items[blk] += 1
print('rows', total, 'blocks', nblocks, 'active blocks', int((items > 0).sum()), 'items', int(items.sum()))
for t in types:
    a, b = off[t] // 16, (off[t] + node_d[t].numel() + 15) // 16
    act = items[a:b] > 0
    last = int(act.nonzero().max()) if act.any() else -1
    print(f'  {t}: blocks {a} ... {b}, active {int(act.sum())}, last active at +{last}, items {int(items[a:b].sum())}')
per = torch.zeros(G, dtype=torch.long, device=dev)
per.index_add_(0, torch.arange(nblocks, device=dev) % G, items)
print(f'G = {G}: items per workgroup mean {per.float().mean():.2f} max {int(per.max())} min {int(per.min())}; '
      f'histogram {torch.bincount(per).tolist()}')
# dense assignment of the active blocks
act = items[items > 0]
per2 = torch.zeros(G, dtype=torch.long, device=dev)
per2.index_add_(0, torch.arange(act.numel(), device=dev) % G, act)
print(f'active blocks dealt densely: max {int(per2.max())} min {int(per2.min())}; histogram {torch.bincount(per2).tolist()}')
# greedy by ticket (next free workgroup takes the next active block): approximated by sorting
