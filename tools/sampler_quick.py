"""Quick C3 sampler timing (no CPU baseline): python tools/sampler_quick.py [batches]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_sampler
print(json.dumps(bench_sampler.run(torch.device('cuda:0'), batches=int(sys.argv[1]) if len(sys.argv) > 1 else 40,
                                   cpu_batches=0)))
