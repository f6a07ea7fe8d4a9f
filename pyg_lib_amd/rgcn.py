"""One relational graph convolution over a sampled heterogeneous neighbourhood, on the device.

SURVEY.md 8(f) N1 / BASELINE.json configs[4]: the step either side of the two hot operators --

    hetero_neighbor_sample  ->  gather_coo (neighbour features)  ->  segment_matmul (one weight per
    relation)  ->  scatter_sum (into the expanded nodes)

The sampler already emits its edges grouped by relation, so the relation pointer of
``segment_matmul`` is just the running sum of the per-relation edge counts (tensor sizes, known
on the host without a synchronisation) and no ``index_sort`` by relation is needed in between.
Node features of all types live in ONE ``[sum_t n_t, F]`` buffer (type offsets), so one gather and
one scatter serve every relation.

For edge type ``(src, rel, dst)`` ``row`` always holds local ids of ``src``-type nodes and ``col`` local ids of
``dst``-type nodes; what ``csc`` changes is which end was EXPANDED and which one SAMPLED
(pyg_lib/csrc/sampler/cpu/neighbor_kernel.cpp:715-719 picks the expanded type, :147-159 swaps the two vectors on return):

    csc=False  the graph is CSR over ``src``: ``row`` = the expanded ``src`` nodes (nondecreasing), ``col`` = their sampled
               ``dst`` neighbours; messages flow col -> row:   out[row + off[src]] += x[col + off[dst]] @ W_r
    csc=True   the graph is CSC over ``dst`` (the reference's own MAG benchmark, benchmark/sampler/hetero_neighbor.py:106-124,
               and PyG's loaders): ``col`` = the expanded ``dst`` nodes (nondecreasing), ``row`` = their sampled ``src``
               neighbours; messages flow row -> col:           out[col + off[dst]] += x[row + off[src]] @ W_r

so the type offsets never swap -- the index ROLES (gather / scatter) do (:func:`edge_roles`).
"""
import os
import weakref
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import ops

EdgeType = Tuple[str, str, str]


def pending_index_error() -> int:
    r"""The fused layer validates its gather / scatter indices on the device WITHOUT synchronising (a stale ``node_id`` is
    redirected to row 0 instead of reading out of bounds; ``PYG_HIP_RGCN_CHECK=1``: synchronising check, ``=0``: none).
    Returns and clears what the launches so far have found on the current device: 0 nothing, 1 a gather index, 2 a scatter
    index out of range, 3 scatter indices that were promised grouped (``grouped=True``) and are not (meaningful after ``torch.cuda.synchronize()``); the next fused call raises for it otherwise.

    The word is ONE per device (pinned host memory, allocated by the first fused call of the process on that device --
    make that call outside a HIP-graph capture, e.g. in the warm-up): an error of one call is reported by whichever
    fused call or ``pending_index_error()`` comes next on the device, from any thread or model.  Where errors must be
    attributed to their call, use ``PYG_HIP_RGCN_CHECK=1`` (synchronising, fails in the call that has the bad index)."""
    from . import _capi
    return int(_capi.lib().pyg_hip_rgcn_pending_error())


_last_path = ['none']


def last_layer_path() -> str:
    r"""Which implementation the last layer call of this process took: ``'chain'`` (gather_coo -> segment_matmul ->
    scatter_sum), ``'atomic'`` (``pyg::rgcn_fused`` with packed atomics) or ``'grouped'`` (the atomic-free kernel).
    Diagnostics, like ``sampler.last_mode``."""
    return _last_path[0]


def type_offsets(num_nodes: Dict[str, int], node_types: List[str]) -> Dict[str, int]:
    off, acc = {}, 0
    for t in node_types:
        off[t] = acc
        acc += int(num_nodes[t])
    off['__total__'] = acc
    return off


def edge_roles(et: EdgeType, row_dict: Dict[EdgeType, Tensor], col_dict: Dict[EdgeType, Tensor], csc: bool):
    r"""``(gather index, gather node type, scatter index, scatter node type)`` of one relation's sampled edges: the
    scatter index is the vector of EXPANDED nodes (``row`` / src for ``csc=False``, ``col`` / dst for ``csc=True``), the
    gather index the vector of sampled neighbours."""
    src, _, dst = et
    if csc:
        return row_dict[et], src, col_dict[et], dst
    return col_dict[et], dst, row_dict[et], src


def out_offsets(offsets: Dict[str, int], num_out_rows: Optional[Dict[str, int]]) -> Dict[str, int]:
    r"""Row layout of the layer's OUTPUT.  ``None``: the layout of `x` (every sampled node gets a row).  Otherwise the
    reference's ``dim_size`` (pyg_lib/csrc/ops/scatter.cpp:156-160: the caller sizes the reduction's output), per node
    type: type ``t`` gets ``num_out_rows[t]`` rows (types not named: none) in the order of `offsets`, and rows behind them
    are neither computed nor zero-filled -- a layer only needs the nodes that were EXPANDED (they come first in every
    type's list: ``sum(num_sampled_nodes_per_hop[t][:-1])``), the last hop's discoveries receive nothing."""
    if num_out_rows is None:
        return offsets
    types = sorted((t for t in offsets if t != '__total__'), key=lambda t: offsets[t])
    for t in num_out_rows:
        if t not in offsets:
            raise ValueError(f"num_out_rows names an unknown node type '{t}'")
    return type_offsets({t: int(num_out_rows.get(t, 0)) for t in types}, types)


def _scatter_rows(ooff: Dict[str, int], roles, edge_types: List[EdgeType]) -> List[int]:
    # rows of every relation's destination type = the bound of its scatter index (pyg_hip_rgcn_relation::scatter_rows; 0
    # would mean "up to the end of out" there: a relation WITH edges into a type without rows is refused here)
    trows = _type_rows(ooff)
    srows = [trows[r[3]] for r in roles]
    for et, r, n in zip(edge_types, roles, srows):
        if n == 0 and r[2].numel() > 0:
            raise RuntimeError(f"rgcn_layer: {et} has edges into node type '{r[3]}', which has no output rows (num_out_rows)")
    return srows


def _type_rows(offsets: Dict[str, int]) -> Dict[str, int]:
    # rows of every node type in an offsets dict (types in offset order; the last one ends at '__total__')
    types = sorted((t for t in offsets if t != '__total__'), key=lambda t: offsets[t])
    ends = [offsets[t] for t in types[1:]] + [offsets['__total__']]
    return {t: e - offsets[t] for t, e in zip(types, ends)}


class _GatherRows(torch.autograd.Function):
    r"""``x[index]`` for an UNSORTED index: forward = the ``gather_coo`` kernel, backward = ``scatter_sum`` of the
    gradient.  ``pyg::gather_coo``'s own autograd formula is ``segment_sum_coo`` -- the reference's
    (ops/autograd/segment_coo_kernel.cpp) -- which is only valid for the SORTED index its contract asks for; the
    neighbour columns of a sampled relation are not sorted."""

    @staticmethod
    def forward(ctx, x, index):
        ctx.save_for_backward(index)
        ctx.rows = x.size(0)
        return ops.gather_coo(x, index)

    @staticmethod
    def backward(ctx, grad_out):
        (index,) = ctx.saved_tensors
        return ops.scatter_sum(grad_out.contiguous(), index, 0, None, ctx.rows), None


def rgcn_layer(x: Tensor, offsets: Dict[str, int], row_dict: Dict[EdgeType, Tensor],
               col_dict: Dict[EdgeType, Tensor], edge_types: List[EdgeType], weight: Tensor,
               csc: bool = False, num_out_rows: Optional[Dict[str, int]] = None) -> Tensor:
    r"""out[row] += x[col] @ weight[r] (``csc=True``: out[col] += x[row] @ weight[r]) over every sampled edge of every
    relation r -- messages flow from the sampled neighbours to the nodes they were sampled for.

    Args:
        x: ``[sum_t n_t, F_in]`` features of the sampled nodes, types concatenated at `offsets`.
        offsets: first row of every node type in `x` (``type_offsets``).
        row_dict, col_dict: local indices from ``hetero_neighbor_sample`` called with the same `csc`.
        edge_types: relation order; ``weight[i]`` belongs to ``edge_types[i]``.
        weight: ``[R, F_in, F_out]``.
        num_out_rows: rows of the output per node type (:func:`out_offsets`; a scatter index at or behind its type's
            count is an error, as a ``dim_size`` that is too small is for ``scatter_sum``).
    Returns:
        ``[sum_t n_t, F_out]`` aggregated messages (same type layout as `x`), or ``[sum_t num_out_rows[t], F_out]``.
    """
    _last_path[0] = 'chain'
    ooff = out_offsets(offsets, num_out_rows)
    counts, gather_idx, scatter_idx = [0], [], []
    for et in edge_types:
        g, g_t, s, s_t = edge_roles(et, row_dict, col_dict, csc)
        counts.append(counts[-1] + g.numel())
        gather_idx.append(g + offsets[g_t] if offsets[g_t] else g)
        if num_out_rows is not None and s.numel() and not (x.is_cuda and torch.cuda.is_current_stream_capturing()):
            # (the chain's scatter would otherwise write another type's rows; the fused kernels check on the device)
            if int(s.max()) >= int(num_out_rows.get(s_t, 0)):
                raise RuntimeError(f"rgcn_layer: a scatter index of {et} is not below num_out_rows['{s_t}']")
        scatter_idx.append(s + ooff[s_t] if ooff[s_t] else s)
    total = ooff['__total__']
    if counts[-1] == 0:
        return x.new_zeros(total, weight.size(-1))
    gidx = torch.cat(gather_idx)
    sidx = torch.cat(scatter_idx)
    ptr = torch.tensor(counts, dtype=torch.long)  # host pointer: staged, never synchronises
    feats = _GatherRows.apply(x, gidx)                        # [E, F_in]
    msgs = ops.segment_matmul(feats, ptr, weight)              # [E, F_out]
    return ops.scatter_sum(msgs, sidx, dim=0, dim_size=total)  # [sum_t n_t, F_out]


# dX of the fused layer: 'atomic' (the fused kernel with swapped roles and packed atomics: fastest, sums in arrival order),
# 'grouped' (the atomic-free kernel on the transposed sample: the same bits on every run; C5: 2.1 x the atomic kernel's
# time -- the transposed sample has ~1.3 edges per (row, relation), so W is fetched per handful of edges), 'auto' (default):
# 'grouped' under torch.use_deterministic_algorithms(True), 'atomic' otherwise.  PYG_RGCN_DX / set_dx_mode().
_DX_MODE = [os.environ.get('PYG_RGCN_DX', 'auto')]


def set_dx_mode(mode: str) -> str:
    r"""``'auto'`` | ``'grouped'`` | ``'atomic'`` (see above); returns the previous mode."""
    if mode not in ('auto', 'grouped', 'atomic'):
        raise ValueError("dX mode must be 'auto', 'grouped' or 'atomic'")
    before, _DX_MODE[0] = _DX_MODE[0], mode
    return before


def _dx_grouped() -> bool:
    m = _DX_MODE[0]
    return m == 'grouped' or (m != 'atomic' and torch.are_deterministic_algorithms_enabled())


_GROUPED_MAX_RELATIONS = 512   # kGroupedMaxRel of csrc/hip/rgcn_grouped.h: the relations' row ranges live in LDS


def _resolve_grouped(grouped: Optional[bool], scatter: List[Tensor]) -> bool:
    # None: yes if every scatter vector is an expanded-node output of this package's samplers (`row` for csc=False, `col`
    # for csc=True: nondecreasing by construction) that has not been written to since
    if grouped is None:
        from . import sampler
        grouped = all(sampler.rows_are_grouped(s) for s in scatter)
    return bool(grouped) and len(scatter) <= _GROUPED_MAX_RELATIONS


def _fusable(x: Tensor, weight: Tensor, grouped: bool = False) -> bool:
    # the atomic kernel: 16-bit, F_in = F_out = 128; the grouped (atomic-free) kernel: 16-bit with F_in, F_out any multiples
    # of 8 up to 256, or float32 with multiples of 4 up to 128
    if x.dim() != 2 or weight.dim() != 3:
        return False
    K, M = x.size(1), weight.size(2)
    if not grouped:
        shape_ok = K == 128 and M == 128
    elif x.dtype == torch.float32:   # rows of multiples of 16 bytes up to 512
        shape_ok = 4 <= K <= 128 and 4 <= M <= 128 and K % 4 == 0 and M % 4 == 0
    else:
        shape_ok = 8 <= K <= 256 and 8 <= M <= 256 and K % 8 == 0 and M % 8 == 0
    dtypes = (torch.bfloat16, torch.float16, torch.float32) if grouped else (torch.bfloat16, torch.float16)
    return (x.is_cuda and x.dtype in dtypes and shape_ok and weight.size(1) == K and weight.dtype == x.dtype and
            weight.device == x.device)


def _fresh_out(like: Tensor, rows: int, cols: int, grouped: bool) -> Tensor:
    # the grouped kernel writes every row once; the atomic kernel accumulates into zeros
    return like.new_empty(rows, cols) if grouped else like.new_zeros(rows, cols)


_flat_indices: Dict[Tuple, Tuple] = {}


def _rel_ptr_and_indices(gather: List[Tensor], scatter: List[Tensor], goff: List[int], soff: List[int]):
    # (relation pointer, concatenated global gather / scatter indices) of a sample: what the weight gradient and the chain
    # need; remembered per sample like the transposed sample (identity + version, weakly) -- 2 R + 2 small launches that
    # every layer of a model and every step on the same batch would otherwise repeat
    key = tuple((id(t), t._version) for t in gather) + tuple((id(t), t._version) for t in scatter) + tuple(goff) + tuple(soff)
    hit = _flat_indices.get(key)
    if hit is not None and all(r() is t for r, t in zip(hit[0], list(gather) + list(scatter))):
        return hit[1]
    counts = [0]
    for g in gather:
        counts.append(counts[-1] + g.numel())
    gidx = torch.cat([g + o if o else g for g, o in zip(gather, goff)])
    sidx = torch.cat([s + o if o else s for s, o in zip(scatter, soff)])
    val = (torch.tensor(counts, dtype=torch.long), gidx, sidx)
    if len(_flat_indices) >= 8:
        _flat_indices.pop(next(iter(_flat_indices)))
    _flat_indices[key] = ([weakref.ref(t) for t in list(gather) + list(scatter)], val)
    return val


# ---- the transposed sample: every relation's edges grouped by their SOURCE row -------------------------------------------
# The backward of the layer scatters through the forward's gather index (dX[g_e] += dOut[s_e] @ W_r^T), which is not grouped:
# sampled neighbours come in draw order.  ONE stable sort of all relations' edges by (relation, gather index) per sample makes
# it so -- then the atomic-free owner-computes kernel serves the backward too (no float atomics anywhere in training, the same
# bits on every run).  The sort depends on the sample only, not on the layer: it is remembered for the tensors it was
# made from (identity + version, weakly -- like sampler.rows_are_grouped), so the layers of a model share it.
_transposed: Dict[Tuple, Tuple] = {}


def _transposed_sample(gather: List[Tensor], scatter: List[Tensor], gather_rows: int):
    key = tuple((id(t), t._version) for t in gather) + tuple((id(t), t._version) for t in scatter)
    hit = _transposed.get(key)
    if hit is not None and all(r() is t for r, t in zip(hit[0], list(gather) + list(scatter))):
        return hit[1], hit[2]
    counts = [g.numel() for g in gather]
    R = len(gather)
    # key = relation * rows + gather index (< rows): relations stay in list order, edges of a relation come out grouped by
    # source row and, inside a row, in edge order (stable)
    keys = torch.cat([g + r * gather_rows if r else g for r, g in enumerate(gather)])
    skeys, perm = ops.index_sort(keys, max_value=max(R * gather_rows, 1))
    g_sorted = list(torch.split(skeys % gather_rows if R > 1 else skeys, counts))
    s_perm = list(torch.split(torch.cat(scatter)[perm], counts))
    if len(_transposed) >= 8:   # a handful of live samples at most
        _transposed.pop(next(iter(_transposed)))
    refs = [weakref.ref(t) for t in list(gather) + list(scatter)]
    _transposed[key] = (refs, g_sorted, s_perm)
    return g_sorted, s_perm


def _dx_scatter(grad_out: Tensor, weight: Tensor, gather: List[Tensor], scatter: List[Tensor], goff: List[int],
                soff: List[int], rows: int, grows: Optional[List[int]] = None) -> Tensor:
    r"""dX[g_e] += dOut[s_e] @ W_r^T.  Shapes the atomic-free kernel takes: on the TRANSPOSED sample (edges grouped by source
    row, :func:`_transposed_sample`) with the two index roles swapped and W_r^T as weight -- no atomics, bit-reproducible.
    That is the path under ``torch.use_deterministic_algorithms(True)`` and with ``set_dx_mode('grouped')``; otherwise the
    atomic fused kernel with swapped roles (16-bit, 128 x 128), and for the remaining shapes the chain gather ->
    segment_matmul -> scatter_sum (atomic-free in deterministic mode: stable sort + CSR rows)."""
    wt = weight.transpose(1, 2).contiguous()
    E = sum(g.numel() for g in gather)
    if E == 0:
        return grad_out.new_zeros(rows, weight.size(1))
    if _dx_grouped() and _fusable(grad_out, wt, True) and len(gather) <= _GROUPED_MAX_RELATIONS:
        g_sorted, s_perm = _transposed_sample(gather, scatter, rows)
        gx = grad_out.new_empty(rows, weight.size(1))   # (written once per row by the grouped kernel)
        torch.ops.pyg.rgcn_fused(grad_out, s_perm, g_sorted, soff, goff, wt, gx, True, grows)
        return gx
    if torch.are_deterministic_algorithms_enabled() or weight.size(1) != 128 or weight.size(2) != 128 or \
            weight.dtype == torch.float32:   # (the atomic kernel: 16-bit, 128 x 128)
        ptr, gidx, sidx = _rel_ptr_and_indices(gather, scatter, goff, soff)
        msgs = ops.segment_matmul(ops.gather_coo(grad_out, sidx), ptr, wt)   # (the kernel takes any index order)
        return ops.scatter_sum(msgs, gidx, dim=0, dim_size=rows)
    gx = grad_out.new_zeros(rows, weight.size(1))   # (contiguous whatever x's strides are: the kernel accumulates into it in place)
    torch.ops.pyg.rgcn_fused(grad_out, scatter, gather, soff, goff, wt, gx)
    return gx


class _RGCNFused(torch.autograd.Function):
    r"""out = rgcn_fused(x, W) with gradients.  Per edge e of relation r:  out[s_e] += x[g_e] @ W_r, hence

        dX[g_e] += dOut[s_e] @ W_r^T     -- the SAME fused kernel with the two index vectors swapped and W_r^T as weight
                                            (gather dOut rows, multiply, scatter-add into dX);
        dW_r     = sum_e x[g_e]^T dOut[s_e] = X_r^T dY_r over the relation's gathered rows -- the weight-gradient
                   kernel (``pyg::segment_matmul_grad_other``: one launch for all relations) on the two gathers.

    This is what autograd derives for the reference's chain gather_coo -> segment_matmul -> scatter_sum
    (ops/autograd/segment_coo_kernel.cpp GatherCOO, ops/autograd/matmul_kernel.cpp:68-111,
    ops/autograd/scatter_kernel.cpp ScatterSum) with the [E, F] intermediates of the dX path never materialised."""

    @staticmethod
    def forward(ctx, x, weight, total, goff, soff, grouped, srows, grows, *index):
        R = len(index) // 2
        gather, scatter = list(index[:R]), list(index[R:])
        out = _fresh_out(x, total, weight.size(-1), grouped)
        torch.ops.pyg.rgcn_fused(x, gather, scatter, goff, soff, weight, out, grouped, srows)
        ctx.save_for_backward(x, weight, *index)
        ctx.meta = (goff, soff, R, grows)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors[:2]
        index = ctx.saved_tensors[2:]
        goff, soff, R, grows = ctx.meta
        gather, scatter = list(index[:R]), list(index[R:])
        grad_out = grad_out.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _dx_scatter(grad_out, weight, gather, scatter, goff, soff, x.size(0), grows)
        if ctx.needs_input_grad[1]:
            ptr, gidx, sidx = _rel_ptr_and_indices(gather, scatter, goff, soff)
            if gidx.numel() == 0:
                gw = torch.zeros_like(weight)
            else:
                gw = torch.ops.pyg.segment_matmul_grad_other(ops.gather_coo(x, gidx), ptr, ops.gather_coo(grad_out, sidx))
        return (gx, gw, None, None, None, None, None, None) + (None,) * len(index)


class _RGCNFusedTables(torch.autograd.Function):
    r"""``rgcn_fused_tables`` with gradients: the weight gradient as in :class:`_RGCNFused` (the relation's source rows
    are gathered from the tables through ``node_id`` for it); feature tables that require a gradient receive
    ``index_add`` of the per-batch dX (computed by the fused kernel with swapped roles), exactly what autograd gives the
    fallback ``cat([feat[t][node_id[t]]])``."""

    @staticmethod
    def forward(ctx, weight, T, gtype, soff, grouped, out_rows, srows, *tensors):
        feat, node_id = list(tensors[:T]), list(tensors[T:2 * T])
        index = tensors[2 * T:]
        R = len(index) // 2
        gather, scatter = list(index[:R]), list(index[R:])
        n_t = [t.numel() for t in node_id]
        out = _fresh_out(feat[0], out_rows, weight.size(-1), grouped)
        torch.ops.pyg.rgcn_fused_tables(feat, node_id, gtype, gather, scatter, soff, weight, out, grouped, srows)
        ctx.save_for_backward(weight, *tensors)
        ctx.meta = (T, gtype, soff, R, n_t)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        weight = ctx.saved_tensors[0]
        tensors = ctx.saved_tensors[1:]
        T, gtype, soff, R, n_t = ctx.meta
        feat, node_id = list(tensors[:T]), list(tensors[T:2 * T])
        index = tensors[2 * T:]
        gather, scatter = list(index[:R]), list(index[R:])
        grad_out = grad_out.contiguous()
        toff = [0]
        for n in n_t:
            toff.append(toff[-1] + n)
        goff = [toff[t] for t in gtype]   # the per-batch matrix cat([feat[t][node_id[t]]]) the tables stand for
        gw = None
        gfeat = [None] * T
        if ctx.needs_input_grad[0]:
            ptr, gidx, sidx = _rel_ptr_and_indices(gather, scatter, goff, soff)
            if gidx.numel() == 0:
                gw = torch.zeros_like(weight)
            else:
                xb = torch.cat([f[n] for f, n in zip(feat, node_id)])   # the per-batch matrix the tables stand for
                gw = torch.ops.pyg.segment_matmul_grad_other(ops.gather_coo(xb, gidx), ptr, ops.gather_coo(grad_out, sidx))
        if any(ctx.needs_input_grad[7 + t] for t in range(T)):
            gx = _dx_scatter(grad_out, weight, gather, scatter, goff, soff, toff[-1], [max(n_t[t], 1) for t in gtype])
            for t in range(T):
                if ctx.needs_input_grad[7 + t]:
                    gfeat[t] = torch.zeros_like(feat[t]).index_add_(0, node_id[t], gx[toff[t]:toff[t + 1]])
        return (gw, None, None, None, None, None, None) + tuple(gfeat) + (None,) * (len(tensors) - T)


def rgcn_layer_fused(x: Tensor, offsets: Dict[str, int], row_dict: Dict[EdgeType, Tensor],
                     col_dict: Dict[EdgeType, Tensor], edge_types: List[EdgeType], weight: Tensor,
                     csc: bool = False, grouped: Optional[bool] = None,
                     num_out_rows: Optional[Dict[str, int]] = None) -> Tensor:
    r"""Same result as :func:`rgcn_layer` from ONE launch (``pyg::rgcn_fused``, csrc/hip/rgcn.hip): source rows are
    gathered straight into the matmul's operand tile, messages are summed per destination run inside the workgroup
    and added with packed atomics -- neither ``feats`` nor ``msgs`` exist in HBM, and the sampler's per-relation
    index vectors are read in place (no ``torch.cat``).  16-bit features with ``F_in = F_out = 128`` (``grouped=True``:
    any multiples of 8 up to 256, or float32 with multiples of 4 up to 128); anything else takes the three-op chain.

    Differentiable: with gradients recorded for ``x`` or ``weight`` the forward still is the one fused launch, and the
    backward runs the fused kernel with swapped roles for dX (atomic adds; under
    ``torch.use_deterministic_algorithms(True)`` / ``set_dx_mode('grouped')`` the atomic-free kernel on the transposed
    sample: :func:`_dx_scatter`) and the weight-gradient kernel on the gathered rows for dW (:class:`_RGCNFused`).  Accumulation: messages are rounded to the storage type (what the chain materialises),
    summed in fp32 per run of equal destinations inside a 32-edge wave tile and added to ``out`` with one packed
    16-bit atomic per run -- a destination whose edges are split over many runs (many relations, tile boundaries) is
    rounded once per run, where ``scatter_sum`` rounds once per destination.  Under
    ``torch.use_deterministic_algorithms(True)`` the atomic-free chain (:func:`rgcn_layer`) runs instead.

    ``grouped=True`` promises that every relation's SCATTER vector (``row_dict[et]`` for ``csc=False``, ``col_dict[et]``
    for ``csc=True``: :func:`edge_roles`) is NONDECREASING -- true for what ``hetero_neighbor_sample`` /
    ``neighbor_sample`` return (the edges of a relation come grouped by the node they were sampled for).
    The default ``None`` means: yes, if every such vector IS a sampler output (the very tensor objects
    ``pyg_lib_amd.sampler`` returned, not written to since: ``sampler.rows_are_grouped``), so the usual pipeline
    sampler -> layer takes the atomic-free kernel without a flag; copies, slices, tensors modified in place and
    hand-made edge lists take the atomic kernel unless promised.
    Then the forward is the ATOMIC-FREE kernel (``PYG_HIP_RGCN_GROUPED``, csrc/hip/rgcn_grouped.h): a workgroup owns 32
    output rows, sums every row's source features in fp32 in edge order, multiplies the sums of a relation with its
    weight in one MFMA tile and writes each row once -- no zero fill, no atomics, the same bits on every run (also the
    path under ``torch.use_deterministic_algorithms(True)``), feature sums and results rounded once each.  The promise is
    verified on the device like the indices (:func:`pending_index_error` = 3 / ``PYG_HIP_RGCN_CHECK=1``).  It is built for
    SAMPLED neighbourhoods (rows of at most a fan-out of edges per relation): a row is walked 16 edges at a time by its
    workgroup, so on an unsampled power-law graph a destination with 50 000 edges of one relation takes 5 ms by itself --
    leave ``grouped`` at its default there (the atomic kernel: 0.24 ms on the same graph).

    ``num_out_rows`` (the reductions' ``dim_size``, per node type: :func:`out_offsets`): only that many rows per type are
    computed and written -- on a C5 batch 427 k rows exist and 29 k can receive anything; the rest of the full-size
    result is zero rows nobody reads."""
    ooff = out_offsets(offsets, num_out_rows)
    total = ooff['__total__']
    roles = [edge_roles(et, row_dict, col_dict, csc) for et in edge_types]
    grouped = _resolve_grouped(grouped, [r[2] for r in roles])
    # torch.use_deterministic_algorithms(True): the fused kernel adds with packed 16-bit atomics (order-dependent); the
    # three-op chain is atomic-free in that mode (gather, per-relation MFMA tiles, scatter_sum through a stable sort)
    if not _fusable(x, weight, grouped) or (torch.are_deterministic_algorithms_enabled() and not grouped):
        return rgcn_layer(x, offsets, row_dict, col_dict, edge_types, weight, csc, num_out_rows)
    gather, scatter = [r[0] for r in roles], [r[2] for r in roles]
    goff, soff = [offsets[r[1]] for r in roles], [ooff[r[3]] for r in roles]
    srows = _scatter_rows(ooff, roles, edge_types)
    _last_path[0] = 'grouped' if grouped else 'atomic'
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        trows = _type_rows(offsets)
        grows = [max(trows[r[1]], 1) for r in roles]   # rows of every relation's SOURCE type: the bound of the backward's scatter
        return _RGCNFused.apply(x, weight, total, goff, soff, grouped, srows, grows, *gather, *scatter)
    out = _fresh_out(x, total, weight.size(-1), grouped)
    return torch.ops.pyg.rgcn_fused(x, gather, scatter, goff, soff, weight, out, grouped, srows)


def rgcn_layer_fused_tables(feat_dict: Dict[str, Tensor], node_id_dict: Dict[str, Tensor], node_types: List[str],
                            row_dict: Dict[EdgeType, Tensor], col_dict: Dict[EdgeType, Tensor],
                            edge_types: List[EdgeType], weight: Tensor, csc: bool = False,
                            grouped: Optional[bool] = None, num_out_rows: Optional[Dict[str, int]] = None) -> Tensor:
    r"""The fused layer straight from the GLOBAL feature tables: what

        x = torch.cat([feat_dict[t][node_id_dict[t]] for t in node_types])
        rgcn_layer_fused(x, type_offsets(...), row_dict, col_dict, edge_types, weight)

    computes, without ``x``: every relation's rows are gathered through the sampler's ``node_id`` of its source type
    inside the kernel (``pyg::rgcn_fused_tables``), so the per-batch feature matrix, the ATen gathers and the ``cat``
    disappear.  Returns ``[sum_t len(node_id_dict[t]), F_out]`` in ``node_types`` order.  Same conditions (16-bit,
    ``F = 128``) as :func:`rgcn_layer_fused`, otherwise the chain above runs; differentiable in ``weight`` and in every
    feature table that requires a gradient (:class:`_RGCNFusedTables`).  ``grouped=True``: the atomic-free kernel, see
    :func:`rgcn_layer_fused`; ``num_out_rows``: rows of the output per node type (:func:`out_offsets`)."""
    off = type_offsets({t: node_id_dict[t].numel() for t in node_types}, node_types)
    ooff = out_offsets(off, num_out_rows)
    roles = [edge_roles(et, row_dict, col_dict, csc) for et in edge_types]
    grouped = _resolve_grouped(grouped, [r[2] for r in roles])
    f0 = feat_dict[node_types[0]]
    feats = [feat_dict[t] for t in node_types]
    nids = [node_id_dict[t] for t in node_types]
    needs_grad = torch.is_grad_enabled() and (weight.requires_grad or any(f.requires_grad for f in feats))
    # every table and the weight: one device, one 16-bit type, F = 128 (anything else: the chain, which checks nothing
    # more than its own ops do)
    ok = _fusable(f0, weight, grouped) and all(f.dim() == 2 and f.size(1) == f0.size(1) and f.dtype == f0.dtype and
                                               f.device == f0.device for f in feats) and \
        all(n.device == f0.device and n.dtype == torch.long and n.dim() == 1 for n in nids)
    if not ok or (torch.are_deterministic_algorithms_enabled() and not grouped):   # (deterministic mode: see rgcn_layer_fused)
        x = torch.cat([feat_dict[t][node_id_dict[t]] for t in node_types])
        return rgcn_layer_fused(x, off, row_dict, col_dict, edge_types, weight, csc, grouped, num_out_rows)
    tidx = {t: i for i, t in enumerate(node_types)}
    gather, scatter = [r[0] for r in roles], [r[2] for r in roles]
    gtype, soff = [tidx[r[1]] for r in roles], [ooff[r[3]] for r in roles]
    srows = _scatter_rows(ooff, roles, edge_types)
    _last_path[0] = 'grouped' if grouped else 'atomic'
    if needs_grad:
        return _RGCNFusedTables.apply(weight, len(feats), gtype, soff, grouped, ooff['__total__'], srows, *feats, *nids, *gather, *scatter)
    out = _fresh_out(f0, ooff['__total__'], weight.size(-1), grouped)
    return torch.ops.pyg.rgcn_fused_tables(feats, nids, gtype, gather, scatter, soff, weight, out, grouped, srows)
