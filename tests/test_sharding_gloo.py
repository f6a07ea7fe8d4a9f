"""N > 1 path on CPU: world_size-2 gloo run of the row-sharded segment_matmul driver.

The HIP operator cannot run here, so each rank multiplies its shard with the oracle (passed in as the
`matmul` callable); what is under test is the sharding arithmetic and the all-gather of the outputs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _oracle_matmul(x, ptr, w, bias=None):
    import oracle
    out = oracle.segment_matmul(x.numpy(), ptr.numpy(), w.numpy(), None if bias is None else bias.numpy())
    return torch.from_numpy(out)


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pyg_lib_amd import sharding
        g = torch.Generator().manual_seed(0)
        sizes = torch.tensor([5, 0, 17, 1, 30, 8])
        ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
        n = int(ptr[-1])
        x = torch.randn(n, 16, generator=g)
        w = torch.randn(len(sizes), 16, 8, generator=g)
        r0, r1, lptr = sharding.shard_ptr(ptr, rank, world)
        assert int(lptr[0]) == 0 and int(lptr[-1]) == r1 - r0 and bool((lptr[1:] >= lptr[:-1]).all())
        full = sharding.segment_matmul_sharded(x[r0:r1], ptr, w, rank, world, gather=True, matmul=_oracle_matmul)
        ref = _oracle_matmul(x, ptr, w)
        ok = torch.allclose(full, ref, atol=1e-6) and full.shape == ref.shape
        counts = torch.tensor([r1 - r0])
        gathered = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
        dist.all_gather(gathered, counts)
        ok = ok and int(torch.cat(gathered).sum()) == n
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_row_sharded_segment_matmul_world_size_2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_rows_partition():
    from pyg_lib_amd import sharding
    for n in (0, 1, 7, 21_111_007):
        for w in (1, 2, 3, 8):
            rs = [sharding.shard_rows(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            cnt = sharding.shard_counts(n, w)
            assert sum(cnt) == n and max(cnt) - min(cnt) <= 1
