"""Relation/row sharding of segment_matmul across ranks.

Rows of a segment_matmul are independent, so the relation list is cut at *row* boundaries: rank r
owns the contiguous rows [N*r/W, N*(r+1)/W) and a copy of ``ptr`` clipped to that range (relations
that straddle a cut simply appear, shortened, on both sides).  Every rank then runs the ordinary
single-GPU operator on its shard -- no collective on the data path.  The only exchange step the
path defines is the optional all-gather of the per-rank outputs over RCCL (xGMI), which is
link-bound and therefore kept out of the compute step (callers that consume sharded rows skip it).
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor


def shard_rows(num_rows: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced row range of `rank`."""
    return num_rows * rank // world_size, num_rows * (rank + 1) // world_size


def shard_ptr(ptr: Tensor, rank: int, world_size: int) -> Tuple[int, int, Tensor]:
    """Returns (row_begin, row_end, local_ptr): `ptr` clipped to this rank's rows and rebased to 0.
    `local_ptr` keeps all B+1 entries so that `other[b]` still lines up with segment b."""
    n = int(ptr[-1])
    r0, r1 = shard_rows(n, rank, world_size)
    return r0, r1, ptr.clamp(r0, r1) - r0


def shard_counts(num_rows: int, world_size: int) -> List[int]:
    return [shard_rows(num_rows, r, world_size)[1] - shard_rows(num_rows, r, world_size)[0]
            for r in range(world_size)]


def all_gather_rows(local_out: Tensor, num_rows: int, group=None) -> Tensor:
    """All-gather(v) of the per-rank output rows into the full [N, M] result (RCCL on GPUs, gloo in
    the CPU tests).  Shards differ by at most one row, so each rank contributes a buffer padded to the
    largest shard and one equal-sized all-gather moves everything."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = shard_counts(num_rows, world)
    width = max(counts)
    tail = tuple(local_out.shape[1:])
    send = local_out.contiguous()
    if send.size(0) != width:
        send = torch.cat([send, send.new_zeros((width - send.size(0),) + tail)], dim=0)
    recv = [local_out.new_empty((width,) + tail) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    return torch.cat([r[:c] for r, c in zip(recv, counts)], dim=0)


def segment_matmul_sharded(inputs_local: Tensor, ptr: Tensor, other: Tensor, rank: int, world_size: int,
                           bias: Optional[Tensor] = None, gather: bool = False, group=None,
                           matmul=None) -> Tensor:
    """segment_matmul on this rank's row shard.  `inputs_local` holds rows [r0, r1) of the global input,
    `ptr` is the GLOBAL segment pointer.  With gather=True the full output is assembled on every rank."""
    if matmul is None:
        from pyg_lib_amd import ops
        matmul = ops.segment_matmul
    r0, r1, lptr = shard_ptr(ptr, rank, world_size)
    assert inputs_local.size(0) == r1 - r0, 'inputs_local must hold exactly this rank\'s rows'
    out = matmul(inputs_local, lptr, other, bias) if bias is not None else matmul(inputs_local, lptr, other)
    if gather and world_size > 1:
        return all_gather_rows(out, int(ptr[-1]), group)
    return out
