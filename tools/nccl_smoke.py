"""RCCL smoke on a one-GPU box: a world of ONE rank on the `nccl` backend runs the collectives the sharded matmul
drivers use (all_gather_into_tensor in place, all_reduce, barrier).  Two ranks cannot share one GPU under RCCL, so
the N > 1 code path of bench.py is exercised separately with `--debug-one-device` (gloo); this script shows that
RCCL itself loads and executes on the box.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/nccl_smoke.py
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', device_id=dev)
    from pyg_lib_amd import ops, sharding
    rows = [300, 0, 129, 512, 7]
    xs = [torch.randn(r, 256, device=dev).bfloat16() for r in rows]
    ws = [(torch.randn(256, 256, device=dev) / 16).bfloat16() for _ in rows]
    plan = sharding.GroupPlan(rows, 1)
    outs, pool = sharding.grouped_matmul_sharded(xs, ws, plan, 0, gather=True)
    ref = ops.grouped_matmul(xs, ws)
    assert all(torch.equal(a, b) for a, b in zip(outs, ref))
    buf = torch.zeros(1, 1000, 128, device=dev, dtype=torch.bfloat16)
    buf[0].normal_()
    keep = buf.clone()
    dist.all_gather_into_tensor(buf.view(1000, 128), buf[0])   # in place, as grouped_matmul_sharded does
    full = sharding.all_gather_rows(keep[0], 1000)
    t = torch.ones(4, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    torch.cuda.synchronize()
    assert torch.equal(buf, keep) and torch.equal(full, keep[0]) and t.tolist() == [1.0] * 4
    print('nccl (RCCL) smoke ok: backend', dist.get_backend(), 'world', dist.get_world_size(),
          'nccl version', torch.cuda.nccl.version())
    dist.destroy_process_group()
