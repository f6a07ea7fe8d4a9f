"""How much do independent C3 batches overlap on one MI355X?  K host threads, each with its own HIP stream, run
neighbor_sample back to back (the generator is shared: results are not reproducible here, throughput only).
    python tools/sampler_concurrency.py [batches per thread]          (GPU_MAX_HW_QUEUES=8 ... to vary the queue count)"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler  # noqa: E402
from pyg_lib_amd import sampler  # noqa: E402

if __name__ == '__main__':
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device('cuda:0')
    rowptr, col = bench_sampler.make_graph(dev)
    g = torch.Generator(device='cpu').manual_seed(1)
    seeds = torch.randperm(bench_sampler.N_NODES, generator=g)[:bench_sampler.BATCH * 8 * (per + 3)].to(dev).view(8, per + 3, -1)
    torch.manual_seed(12345)
    for b in range(3):
        sampler.neighbor_sample(rowptr, col, seeds[0, b], bench_sampler.FANOUT)
    torch.cuda.synchronize()
    print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
    base = None
    for K in (1, 2, 4, 8):
        edges = [0] * K
        streams = [torch.cuda.Stream() for _ in range(K)]

        def work(k):
            with torch.cuda.stream(streams[k]):
                for b in range(per):
                    out = sampler.neighbor_sample(rowptr, col, seeds[k, 3 + b], bench_sampler.FANOUT)
                    edges[k] += sum(out[5])
        for rep in range(2):
            edges = [0] * K
            th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        rate = sum(edges) / dt
        base = base or rate
        print(f'K={K}: {dt / (K * per) * 1e3:.3f} ms per batch, {rate / 1e9:.2f} G edges/s, x{rate / base:.2f}', flush=True)
