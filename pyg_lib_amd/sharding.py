"""Relation/row sharding of segment_matmul / grouped_matmul across ranks.

Rows of a segment_matmul are independent, so the relation list is cut at *row* boundaries: rank r
owns the contiguous rows [N*r/W, N*(r+1)/W) and a copy of ``ptr`` clipped to that range (relations
that straddle a cut simply appear, shortened, on both sides).  Every rank then runs the ordinary
single-GPU operator on its shard -- no collective on the data path.  The only exchange step the
path defines is the optional all-gather of the per-rank outputs over RCCL (xGMI), which is
link-bound and therefore kept out of the compute step (callers that consume sharded rows skip it).

``grouped_matmul`` (BASELINE config C4: 512 variable-size groups) shards by GROUPS: whole groups are
bin-packed onto the ranks by their row counts (greedy LPT -- largest group first onto the least loaded
rank; flops and bytes of a group are proportional to its rows when K and M are uniform), every rank runs
one grouped launch over its groups with the outputs written straight into its slot of a preallocated
``[world, max_rows, M]`` pool, and ONE ``all_gather_into_tensor`` (in place: the send buffer is that slot)
completes the pool on every rank.  The per-group results are row slices of the pool -- no list gather,
no concatenation, no padding copies.
"""
from typing import List, Optional, Sequence, Tuple

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
    the CPU tests).  Shards differ by at most one row: when they are all equal, one
    ``all_gather_into_tensor`` lands every shard at its final place in the result; otherwise the result
    buffer carries ``world - 1`` spare rows, every rank's shard lands at ``r * width`` and the (at most
    one-row) gaps are closed by moving the shards down in place, front to back."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = shard_counts(num_rows, world)
    width = max(counts)
    tail = tuple(local_out.shape[1:])
    send = local_out.contiguous()
    full = local_out.new_empty((world * width,) + tail)
    if send.size(0) != width:
        rank = dist.get_rank(group)
        slot = full[rank * width:(rank + 1) * width]
        slot[:send.size(0)].copy_(send)
        send = slot
    dist.all_gather_into_tensor(full, send, group=group)
    if min(counts) == width:
        return full
    pos = 0
    for r, c in enumerate(counts):
        if pos != r * width:
            # destination lies below the source and shards are moved front to back: an overlapping
            # forward move is safe only through a temporary
            full[pos:pos + c].copy_(full[r * width:r * width + c].clone() if r * width - pos < c else
                                    full[r * width:r * width + c])
        pos += c
    return full[:num_rows]


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


# ---------------------------------------------------------------------------------------------------
# grouped_matmul: whole groups per rank (BASELINE config C4)
# ---------------------------------------------------------------------------------------------------

def lpt_assign(rows: Sequence[int], world_size: int) -> List[int]:
    """Greedy longest-processing-time bin packing of groups onto ranks by row count: groups in
    descending size (ties: lower index first) go to the currently least loaded rank (ties: lower rank).
    Returns the owner rank of every group.  Deterministic, so every rank computes the same plan."""
    order = sorted(range(len(rows)), key=lambda i: (-int(rows[i]), i))
    load = [0] * world_size
    owner = [0] * len(rows)
    for i in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += int(rows[i])
    return owner


class GroupPlan:
    """Where every group of a sharded grouped_matmul lives: owner rank, row offset inside the owner's
    slot of the ``[world, max_rows, M]`` pool."""

    def __init__(self, rows: Sequence[int], world_size: int):
        self.rows = [int(r) for r in rows]
        self.world_size = world_size
        self.owner = lpt_assign(self.rows, world_size)
        self.offset = [0] * len(self.rows)
        self.load = [0] * world_size
        for i, r in enumerate(self.owner):  # a rank's groups keep their global order inside its slot
            self.offset[i] = self.load[r]
            self.load[r] += self.rows[i]
        self.max_rows = max(self.load) if self.load else 0

    def local_groups(self, rank: int) -> List[int]:
        return [i for i, r in enumerate(self.owner) if r == rank]

    @property
    def imbalance(self) -> float:
        """max load / mean load (1.0 = perfectly balanced)."""
        mean = sum(self.load) / max(1, self.world_size)
        return self.max_rows / mean if mean > 0 else 1.0


def _grouped_into(inputs: List[Tensor], others: List[Tensor], pool: Tensor) -> List[Tensor]:
    return list(torch.ops.pyg.grouped_matmul_pool(list(inputs), list(others), pool))


def grouped_matmul_sharded(inputs_local: List[Tensor], others_local: List[Tensor], plan: GroupPlan, rank: int,
                           gather: bool = False, group=None, matmul_into=None, out_features: Optional[int] = None,
                           dtype: Optional[torch.dtype] = None, device=None):
    """grouped_matmul over this rank's groups (``plan.local_groups(rank)``, in that order).

    The outputs are written into this rank's slot of a ``[world, max_rows, M]`` pool.  With
    ``gather=False`` returns ``(local_outs, pool)`` -- the local results are row slices of the slot, the
    rest of the pool is untouched; with ``gather=True`` one in-place ``all_gather_into_tensor`` fills the
    other slots and the function returns ``(all_outs, pool)`` with every group's result (global order)
    as a row slice of the pool.  ``matmul_into(inputs, others, slot)`` must write ``inputs[i] @ others[i]``
    to consecutive row ranges of ``slot`` and return them (default: ``pyg::grouped_matmul_pool``).

    A rank that owns no group (more ranks than non-empty groups) cannot read the pool's shape from its tensors:
    pass ``out_features`` / ``dtype`` / ``device`` then; such a rank skips the matmul and still takes part in the
    collective with its (empty) slot."""
    if matmul_into is None:
        matmul_into = _grouped_into
    mine = plan.local_groups(rank)
    assert len(inputs_local) == len(mine) == len(others_local), 'inputs_local must hold exactly this rank\'s groups'
    ref = others_local[0] if others_local else None
    if ref is not None:
        m = ref.size(-1) if out_features is None else int(out_features)
        dtype = ref.dtype if dtype is None else dtype
        device = ref.device if device is None else device
    else:
        if out_features is None or dtype is None or device is None:
            raise ValueError('grouped_matmul_sharded: a rank without groups needs out_features, dtype and device '
                             'to size its slot of the pool')
        m = int(out_features)
    pool = torch.empty((plan.world_size, max(plan.max_rows, 1), m), dtype=dtype, device=device)
    slot = pool[rank]
    local_outs = matmul_into(inputs_local, others_local, slot[:plan.load[rank]]) if mine else []
    if not gather or plan.world_size == 1:
        if not gather:
            return local_outs, pool
    if plan.world_size > 1:
        import torch.distributed as dist
        dist.all_gather_into_tensor(pool.view(plan.world_size * pool.size(1), m), slot, group=group)
    outs = [pool[plan.owner[i], plan.offset[i]:plan.offset[i] + plan.rows[i]] for i in range(len(plan.rows))]
    return outs, pool
