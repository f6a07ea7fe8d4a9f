"""Wall-clock per batch of the HIP sampler on the C3-shaped graph (used under rocprofv3)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler  # noqa: E402

if __name__ == '__main__':
    dev = torch.device('cuda:0')
    r = bench_sampler.run(dev, batches=int(sys.argv[1]) if len(sys.argv) > 1 else 20, warmup=3, cpu_batches=0)
    print(r)
