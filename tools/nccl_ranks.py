"""The real `nccl` (RCCL) path of the sharded matmul drivers with one rank per GPU: run under a launcher,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/nccl_ranks.py

Every rank computes the single-GPU result of the whole job on its own device, then its shard through
`segment_matmul_sharded(gather=True)` / `grouped_matmul_sharded(gather=True)`, and compares BITS: a shard
boundary only cuts the row list, never a row's k order.  Prints one line `nccl ranks ok ...` on rank 0; any
mismatch is a non-zero exit.  (`tests/test_sharding_nccl_gpu.py` runs it when the box has >= 2 GPUs.)
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    assert torch.cuda.device_count() >= world, f'{world} ranks need {world} GPUs, {torch.cuda.device_count()} visible'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    from pyg_lib_amd import ops, sharding
    g = torch.Generator().manual_seed(7)   # the same job on every rank
    # segment_matmul: 23 ragged relations (empty ones included), F = 128 bf16, rows not divisible by the world
    sizes = torch.randint(0, 9000, (23,), generator=g)
    sizes[3] = 0
    sizes[-1] += 1
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    n = int(ptr[-1])
    x = torch.randn(n, 128, generator=g).bfloat16().to(dev)
    w = (torch.randn(len(sizes), 128, 128, generator=g) / 128 ** 0.5).bfloat16().to(dev)
    bias = torch.randn(len(sizes), 128, generator=g).bfloat16().to(dev)
    ref = ops.segment_matmul(x, ptr, w, bias)
    r0, r1 = sharding.shard_rows(n, rank, world)
    full = sharding.segment_matmul_sharded(x[r0:r1], ptr, w, rank, world, bias=bias, gather=True)
    assert full.shape == ref.shape and torch.equal(full.view(torch.int16), ref.view(torch.int16)), 'segment rows differ'
    # grouped_matmul (C4-shaped, small): 37 groups, K = M = 256, LPT-sharded, in-place all-gather of the pool
    rows = [int(v) for v in torch.randint(1, 3000, (37,), generator=g)]
    rows[5] = 0
    xs = [torch.randn(r, 256, generator=g).bfloat16().to(dev) for r in rows]
    ws = [(torch.randn(256, 256, generator=g) / 16).bfloat16().to(dev) for _ in rows]
    refs = ops.grouped_matmul(xs, ws)
    plan = sharding.GroupPlan(rows, world)
    mine = plan.local_groups(rank)
    outs, pool = sharding.grouped_matmul_sharded([xs[i] for i in mine], [ws[i] for i in mine], plan, rank, gather=True,
                                                 out_features=256, dtype=torch.bfloat16, device=dev)
    for i, (a, b) in enumerate(zip(outs, refs)):
        assert a.shape == b.shape and torch.equal(a.view(torch.int16), b.view(torch.int16)), f'group {i} differs'
    ok = torch.ones(1, device=dev)
    dist.all_reduce(ok)
    torch.cuda.synchronize()
    assert int(ok.item()) == world
    if rank == 0:
        print('nccl ranks ok: backend', dist.get_backend(), 'world', world, 'rows', n, 'groups', len(rows),
              'imbalance %.4f' % plan.imbalance, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
