"""C5 (BASELINE.json configs[4]): hetero_neighbor_sample + fused R-GCN layer end to end on a
MAG-shaped synthetic heterogeneous graph: paper 736,389; author 1,134,649;
institution 8,740; field_of_study 59,965; 7 directed relations, ~42 M entries; batch 1024 papers,
fanout [15, 10] (two hops for the layer test), F=128.

    python tools/bench_rgcn.py [--dtype bf16] [--iters 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SIZES = {'paper': 736_389, 'author': 1_134_649, 'institution': 8_740, 'field_of_study': 59_965}
# (src, rel, dst, entries) after ToUndirected: cites 2x5.4M, writes 7.1M (+rev), affiliated 1.0M (+rev),
# has_topic 7.5M (+rev)
RELS = [('paper', 'cites', 'paper', 10_832_542), ('author', 'writes', 'paper', 7_145_660),
        ('paper', 'rev_writes', 'author', 7_145_660), ('author', 'affiliated_with', 'institution', 1_043_998),
        ('institution', 'rev_affiliated_with', 'author', 1_043_998),
        ('paper', 'has_topic', 'field_of_study', 7_505_078), ('field_of_study', 'rev_has_topic', 'paper', 7_505_078)]


def make_graph(device):
    g = torch.Generator(device=device).manual_seed(0)
    rp, cl = {}, {}
    for s, r, d, e in RELS:
        src = torch.randint(0, SIZES[s], (e,), device=device, generator=g)
        deg = torch.bincount(src, minlength=SIZES[s])
        rp[(s, r, d)] = torch.cat([deg.new_zeros(1), deg.cumsum(0)])
        cl[(s, r, d)] = torch.randint(0, SIZES[d], (e,), device=device, generator=g)
    return rp, cl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=1024)
    args = ap.parse_args()
    from pyg_lib_amd import sampler, rgcn
    device = torch.device('cuda', 0)
    dtype = dict(bf16=torch.bfloat16, f32=torch.float32, f16=torch.float16)[args.dtype]
    types = list(SIZES)
    ets = [(s, r, d) for s, r, d, _ in RELS]
    rp, cl = make_graph(device)
    F = 128
    feat = {t: torch.randn(SIZES[t], F, device=device).to(dtype) for t in types}
    W = (torch.randn(len(ets), F, F, device=device) / F ** 0.5).to(dtype)
    fan = {e: [15, 10] for e in ets}
    gs = torch.Generator().manual_seed(1)
    seeds = [torch.randperm(SIZES['paper'], generator=gs)[:args.batch].to(device) for _ in range(args.iters + 3)]

    def one(i):
        torch.manual_seed(100 + i)
        out = sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds[i]}, fan)
        row_d, col_d, node_d = out[0], out[1], out[2]
        off = rgcn.type_offsets({t: node_d[t].numel() for t in types}, types)
        x = torch.cat([feat[t][node_d[t]] for t in types])
        y = rgcn.rgcn_layer(x, off, row_d, col_d, ets, W)
        return out, y

    for i in range(3):
        out, y = one(i)
    torch.cuda.synchronize()
    t_s = t_l = 0.0
    edges = nodes = 0
    t0 = time.perf_counter()
    for i in range(3, 3 + args.iters):
        out, y = one(i)
        edges += sum(v.numel() for v in out[0].values())
        nodes += sum(v.numel() for v in out[2].values())
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / args.iters
    # split: sampler alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3, 3 + args.iters):
        torch.manual_seed(100 + i)
        sampler.hetero_neighbor_sample(rp, cl, {'paper': seeds[i]}, fan)
    torch.cuda.synchronize()
    t_s = (time.perf_counter() - t0) / args.iters
    print(json.dumps(dict(workload='C5 MAG-shaped hetero sample + R-GCN layer', dtype=args.dtype, batch=args.batch,
                          fanout=[15, 10], F=F, ms_total=round(total * 1e3, 3), ms_sampler=round(t_s * 1e3, 3),
                          ms_layer=round((total - t_s) * 1e3, 3), edges_per_batch=edges // args.iters,
                          nodes_per_batch=nodes // args.iters,
                          edges_per_s=round(edges / args.iters / total))))


if __name__ == '__main__':
    main()
