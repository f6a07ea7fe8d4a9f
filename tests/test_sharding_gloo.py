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
    """Rendezvous token of one test: a fresh file for a FileStore (a TCP port picked here could be taken by another pytest
    worker between the probe and the bind: `pytest -n` runs these tests side by side)."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix='pyg_gloo_rdzv_')
    os.close(fd)
    os.unlink(path)
    return path


def _oracle_matmul(x, ptr, w, bias=None):
    import oracle
    out = oracle.segment_matmul(x.numpy(), ptr.numpy(), w.numpy(), None if bias is None else bias.numpy())
    return torch.from_numpy(out)


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    dist.init_process_group('gloo', init_method=f'file://{port}', rank=rank, world_size=world)
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


def _oracle_grouped_into(inputs, others, slot):
    """Stand-in for pyg::grouped_matmul_pool on CPU: every product goes to its row range of `slot`."""
    import oracle
    outs, pos = [], 0
    for a, o in zip(inputs, others):
        r = a.size(0)
        slot[pos:pos + r] = torch.from_numpy(oracle.matmul(a.numpy(), o.numpy())) if r else slot[pos:pos]
        outs.append(slot[pos:pos + r])
        pos += r
    return outs


def _grouped_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    dist.init_process_group('gloo', init_method=f'file://{port}', rank=rank, world_size=world)
    try:
        from pyg_lib_amd import sharding
        g = torch.Generator().manual_seed(0)
        rows = [40, 0, 7, 130, 1, 64, 9, 33, 250, 12, 5]   # every rank regenerates the same global job
        ins = [torch.randn(r, 16, generator=g) for r in rows]
        oth = [torch.randn(16, 8, generator=g) for _ in rows]
        plan = sharding.GroupPlan(rows, world)
        mine = plan.local_groups(rank)
        outs, pool = sharding.grouped_matmul_sharded([ins[i] for i in mine], [oth[i] for i in mine], plan, rank,
                                                     gather=True, matmul_into=_oracle_grouped_into)
        ok = len(outs) == len(rows) and pool.shape == (world, plan.max_rows, 8)
        for i, o in enumerate(outs):
            ok = ok and o.shape == (rows[i], 8) and torch.allclose(o, ins[i] @ oth[i], atol=1e-5)
            # results are row slices of the gathered pool: no copy was made
            ok = ok and (rows[i] == 0 or o.data_ptr() == pool[plan.owner[i], plan.offset[i]].data_ptr())
        # compute-only form: local results only
        louts, _ = sharding.grouped_matmul_sharded([ins[i] for i in mine], [oth[i] for i in mine], plan, rank,
                                                   gather=False, matmul_into=_oracle_grouped_into)
        ok = ok and len(louts) == len(mine) and all(torch.allclose(o, ins[i] @ oth[i], atol=1e-5)
                                                    for o, i in zip(louts, mine))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_group_sharded_grouped_matmul_world_size_2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grouped_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_lpt_plan_is_balanced_and_deterministic():
    import math
    import random
    from pyg_lib_amd import sharding
    random.seed(0)
    rows = [int(math.exp(random.uniform(math.log(256), math.log(65536)))) for _ in range(512)]  # C4's group sizes
    for w in (1, 2, 4, 8):
        p = sharding.GroupPlan(rows, w)
        assert sorted(sum((p.local_groups(r) for r in range(w)), [])) == list(range(512))
        assert sum(p.load) == sum(rows) and p.imbalance < 1.001
        assert p.owner == sharding.GroupPlan(rows, w).owner
        for r in range(w):  # slots are packed: offsets are the running row count of the rank's groups
            pos = 0
            for i in p.local_groups(r):
                assert p.offset[i] == pos
                pos += rows[i]
            assert pos == p.load[r] <= p.max_rows
    assert sharding.lpt_assign([5, 5, 5], 2) == [0, 1, 0]
    assert sharding.lpt_assign([], 3) == []


def test_all_gather_rows_uneven_shards_world_size_2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_uneven_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _uneven_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    dist.init_process_group('gloo', init_method=f'file://{port}', rank=rank, world_size=world)
    try:
        from pyg_lib_amd import sharding
        ok = True
        for n in (7, 8, 1):
            full = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
            r0, r1 = sharding.shard_rows(n, rank, world)
            got = sharding.all_gather_rows(full[r0:r1].clone(), n)
            ok = ok and got.shape == full.shape and torch.equal(got, full)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_shard_rows_partition():
    from pyg_lib_amd import sharding
    for n in (0, 1, 7, 21_111_007):
        for w in (1, 2, 3, 8):
            rs = [sharding.shard_rows(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            cnt = sharding.shard_counts(n, w)
            assert sum(cnt) == n and max(cnt) - min(cnt) <= 1


def _empty_rank_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    dist.init_process_group('gloo', init_method=f'file://{port}', rank=rank, world_size=world)
    try:
        from pyg_lib_amd import sharding
        g = torch.Generator().manual_seed(0)
        rows = [40]   # one group, two ranks: LPT leaves rank 1 without work
        ins = [torch.randn(r, 16, generator=g) for r in rows]
        oth = [torch.randn(16, 8, generator=g) for _ in rows]
        plan = sharding.GroupPlan(rows, world)
        mine = plan.local_groups(rank)
        assert (mine == []) == (rank == 1)
        kw = dict(out_features=8, dtype=torch.float32, device='cpu')
        outs, pool = sharding.grouped_matmul_sharded([ins[i] for i in mine], [oth[i] for i in mine], plan, rank,
                                                     gather=True, matmul_into=_oracle_grouped_into, **kw)
        ok = len(outs) == 1 and torch.allclose(outs[0], ins[0] @ oth[0], atol=1e-5)
        louts, _ = sharding.grouped_matmul_sharded([ins[i] for i in mine], [oth[i] for i in mine], plan, rank,
                                                   gather=False, matmul_into=_oracle_grouped_into, **kw)
        ok = ok and len(louts) == len(mine)
        if rank == 1:  # without the shape arguments the empty rank says what it needs instead of hanging its peers
            try:
                sharding.grouped_matmul_sharded([], [], plan, rank, gather=False, matmul_into=_oracle_grouped_into)
                ok = False
            except ValueError:
                pass
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_rank_without_groups_still_joins_the_all_gather():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_empty_rank_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
