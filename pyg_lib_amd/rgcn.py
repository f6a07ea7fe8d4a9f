"""One relational graph convolution over a sampled heterogeneous neighbourhood, on the device.

SURVEY.md 8(f) N1 / BASELINE.json configs[4]: the step either side of the two hot operators --

    hetero_neighbor_sample  ->  gather_coo (neighbour features)  ->  segment_matmul (one weight per
    relation)  ->  scatter_sum (into the expanded nodes)

The sampler already emits its edges grouped by relation, so the relation pointer of
``segment_matmul`` is just the running sum of the per-relation edge counts (tensor sizes, known
on the host without a synchronisation) and no ``index_sort`` by relation is needed in between.
Node features of all types live in ONE ``[sum_t n_t, F]`` buffer (type offsets), so one gather and
one scatter serve every relation.

For ``csc=False`` sampler output and edge type ``(src, rel, dst)``: ``row`` holds local ids of the
expanded ``src``-type nodes, ``col`` local ids of the sampled ``dst``-type neighbours
(pyg_lib/csrc/sampler/cpu/neighbor_kernel.cpp:587-602); messages flow col -> row.  With
``csc=True`` the roles of the two end types swap (row indexes ``dst``-type nodes).
"""
from typing import Dict, List, Tuple

import torch
from torch import Tensor

from . import ops

EdgeType = Tuple[str, str, str]


def type_offsets(num_nodes: Dict[str, int], node_types: List[str]) -> Dict[str, int]:
    off, acc = {}, 0
    for t in node_types:
        off[t] = acc
        acc += int(num_nodes[t])
    off['__total__'] = acc
    return off


def rgcn_layer(x: Tensor, offsets: Dict[str, int], row_dict: Dict[EdgeType, Tensor],
               col_dict: Dict[EdgeType, Tensor], edge_types: List[EdgeType], weight: Tensor,
               csc: bool = False) -> Tensor:
    r"""out[row] += x[col] @ weight[r] over every sampled edge of every relation r.

    Args:
        x: ``[sum_t n_t, F_in]`` features of the sampled nodes, types concatenated at `offsets`.
        offsets: first row of every node type in `x` (``type_offsets``).
        row_dict, col_dict: local indices from ``hetero_neighbor_sample``.
        edge_types: relation order; ``weight[i]`` belongs to ``edge_types[i]``.
        weight: ``[R, F_in, F_out]``.
    Returns:
        ``[sum_t n_t, F_out]`` aggregated messages (same type layout as `x`).
    """
    counts, gather_idx, scatter_idx = [0], [], []
    for et in edge_types:
        src, _, dst = et
        row_t, col_t = (src, dst) if not csc else (dst, src)
        row, col = row_dict[et], col_dict[et]
        counts.append(counts[-1] + row.numel())
        gather_idx.append(col + offsets[col_t] if offsets[col_t] else col)
        scatter_idx.append(row + offsets[row_t] if offsets[row_t] else row)
    total = offsets['__total__']
    if counts[-1] == 0:
        return x.new_zeros(total, weight.size(-1))
    gidx = torch.cat(gather_idx)
    sidx = torch.cat(scatter_idx)
    ptr = torch.tensor(counts, dtype=torch.long)  # host pointer: staged, never synchronises
    feats = ops.gather_coo(x, gidx)                           # [E, F_in]
    msgs = ops.segment_matmul(feats, ptr, weight)              # [E, F_out]
    return ops.scatter_sum(msgs, sidx, dim=0, dim_size=total)  # [sum_t n_t, F_out]


def rgcn_layer_fused(x: Tensor, offsets: Dict[str, int], row_dict: Dict[EdgeType, Tensor],
                     col_dict: Dict[EdgeType, Tensor], edge_types: List[EdgeType], weight: Tensor,
                     csc: bool = False) -> Tensor:
    r"""Same result as :func:`rgcn_layer` from ONE launch (``pyg::rgcn_fused``, csrc/hip/rgcn.hip): source rows are
    gathered straight into the matmul's operand tile, messages are summed per destination run inside the workgroup
    and added with packed atomics -- neither ``feats`` nor ``msgs`` exist in HBM, and the sampler's per-relation
    index vectors are read in place (no ``torch.cat``).  16-bit features with ``F_in = F_out = 128``; anything else
    takes the three-op chain.

    The fused operator is inference-only (``pyg::rgcn_fused`` has no autograd formula): when gradients are being
    recorded for ``x`` or ``weight`` the differentiable three-op chain runs instead, so a training loop that switches
    to this function keeps learning.  Accumulation: messages are rounded to the storage type (what the chain
    materialises), summed in fp32 per run of equal destinations inside a 32-edge wave tile and added to ``out`` with
    one packed 16-bit atomic per run -- a destination whose edges are split over many runs (many relations, tile
    boundaries) is rounded once per run, where ``scatter_sum`` rounds once per destination."""
    total = offsets['__total__']
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)
    if needs_grad or not (x.dtype in (torch.bfloat16, torch.float16) and x.size(1) == 128 and weight.dim() == 3 and
                          weight.size(-1) == 128 and x.is_cuda and weight.dtype == x.dtype and
                          weight.device == x.device):
        return rgcn_layer(x, offsets, row_dict, col_dict, edge_types, weight, csc)
    gather, scatter, goff, soff = [], [], [], []
    for et in edge_types:
        src, _, dst = et
        row_t, col_t = (src, dst) if not csc else (dst, src)
        gather.append(col_dict[et])
        scatter.append(row_dict[et])
        goff.append(offsets[col_t])
        soff.append(offsets[row_t])
    out = x.new_zeros(total, weight.size(-1))
    return torch.ops.pyg.rgcn_fused(x, gather, scatter, goff, soff, weight, out)


def rgcn_layer_fused_tables(feat_dict: Dict[str, Tensor], node_id_dict: Dict[str, Tensor], node_types: List[str],
                            row_dict: Dict[EdgeType, Tensor], col_dict: Dict[EdgeType, Tensor],
                            edge_types: List[EdgeType], weight: Tensor, csc: bool = False) -> Tensor:
    r"""The fused layer straight from the GLOBAL feature tables: what

        x = torch.cat([feat_dict[t][node_id_dict[t]] for t in node_types])
        rgcn_layer_fused(x, type_offsets(...), row_dict, col_dict, edge_types, weight)

    computes, without ``x``: every relation's rows are gathered through the sampler's ``node_id`` of its source type
    inside the kernel (``pyg::rgcn_fused_tables``), so the per-batch feature matrix, the ATen gathers and the ``cat``
    disappear.  Returns ``[sum_t len(node_id_dict[t]), F_out]`` in ``node_types`` order.  Same conditions (16-bit,
    ``F = 128``, no gradients) as :func:`rgcn_layer_fused`; otherwise the chain above runs."""
    off = type_offsets({t: node_id_dict[t].numel() for t in node_types}, node_types)
    f0 = feat_dict[node_types[0]]
    needs_grad = torch.is_grad_enabled() and (weight.requires_grad or any(f.requires_grad for f in feat_dict.values()))
    # every table and the weight: one device, one 16-bit type, F = 128 (anything else: the chain, which checks nothing
    # more than its own ops do)
    ok = f0.is_cuda and f0.dtype in (torch.bfloat16, torch.float16) and weight.dim() == 3 and weight.size(-1) == 128 and \
        weight.dtype == f0.dtype and weight.device == f0.device and \
        all(f.dim() == 2 and f.size(1) == 128 and f.dtype == f0.dtype and f.device == f0.device
            for f in (feat_dict[t] for t in node_types)) and \
        all(node_id_dict[t].device == f0.device and node_id_dict[t].dtype == torch.long for t in node_types)
    if needs_grad or not ok:
        x = torch.cat([feat_dict[t][node_id_dict[t]] for t in node_types])
        return rgcn_layer_fused(x, off, row_dict, col_dict, edge_types, weight, csc)
    tidx = {t: i for i, t in enumerate(node_types)}
    gather, scatter, gtype, soff = [], [], [], []
    for et in edge_types:
        src, _, dst = et
        row_t, col_t = (src, dst) if not csc else (dst, src)
        gather.append(col_dict[et])
        scatter.append(row_dict[et])
        gtype.append(tidx[col_t])
        soff.append(off[row_t])
    out = f0.new_zeros(off['__total__'], weight.size(-1))
    return torch.ops.pyg.rgcn_fused_tables([feat_dict[t] for t in node_types], [node_id_dict[t] for t in node_types],
                                           gtype, gather, scatter, soff, weight, out)
