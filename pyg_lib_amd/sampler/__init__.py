"""Host-side mirror of ``pyg_lib.sampler`` for the hot path (pyg_lib/sampler/__init__.py:11-200).

Same names, arguments, defaults and return structure as the reference: thin wrappers over
``torch.ops.pyg.neighbor_sample`` / ``torch.ops.pyg.hetero_neighbor_sample``.  The graph
(``rowptr``, ``col``) and the seeds live on a HIP device and the whole multi-hop expansion,
including the first-occurrence-ordered relabelling, runs there (the reference only has a
single-threaded CPU kernel).  Random numbers are drawn from PyTorch's global CPU generator exactly
as the reference's ``RandintEngine`` does, so ``torch.manual_seed(s)`` yields bit-identical samples.
"""
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

NodeType = str
RelType = str
EdgeType = Tuple[str, str, str]


def neighbor_sample(
    rowptr: Tensor,
    col: Tensor,
    seed: Tensor,
    num_neighbors: List[int],
    node_time: Optional[Tensor] = None,
    edge_time: Optional[Tensor] = None,
    seed_time: Optional[Tensor] = None,
    edge_weight: Optional[Tensor] = None,
    csc: bool = False,
    replace: bool = False,
    directed: bool = True,
    disjoint: bool = False,
    temporal_strategy: str = 'uniform',
    return_edge_id: bool = True,
) -> Tuple[Tensor, Tensor, Tensor, Optional[Tensor], List[int], List[int]]:
    r"""Recursively samples neighbors from all node indices in :obj:`seed`
    in the graph given by :obj:`(rowptr, col)` (same contract as
    :func:`pyg_lib.sampler.neighbor_sample`, pyg_lib/sampler/__init__.py:11-100).

    Args:
        rowptr: Compressed source node indices.
        col: Target node indices.
        seed: The seed node indices.
        num_neighbors: The number of neighbors to sample for each node in each
            iteration. If an entry is set to :obj:`-1`, all neighbors will be
            included.
        node_time, edge_time, seed_time: Temporal sampling (requires
            :obj:`disjoint=True`).
        edge_weight: Per-edge float32 / float64 weights for biased sampling
            (without replacement only on the device path).
        csc: If set to :obj:`True`, assumes that the graph is given in CSC
            format :obj:`(colptr, row)`.
        replace: If set to :obj:`True`, will sample with replacement.
        directed: If set to :obj:`False`, will include all edges between all
            sampled nodes (unsupported, as in the reference).
        disjoint: If set to :obj:`True` , will create disjoint subgraphs for
            every seed node.
        temporal_strategy: :obj:`"uniform"` or :obj:`"last"`.
        return_edge_id: If set to :obj:`False`, will not return the indices of
            edges of the original graph.

    Returns:
        Row indices, col indices of the returned subtree/subgraph, original node indices of all
        sampled nodes, optionally the indices of the sampled edges in the original graph, and the
        number of sampled nodes / edges per hop.
    """
    return torch.ops.pyg.neighbor_sample(
        rowptr,
        col,
        seed,
        num_neighbors,
        node_time,
        edge_time,
        seed_time,
        edge_weight,
        csc,
        replace,
        directed,
        disjoint,
        temporal_strategy,
        return_edge_id,
    )


def hetero_neighbor_sample(
    rowptr_dict: Dict[EdgeType, Tensor],
    col_dict: Dict[EdgeType, Tensor],
    seed_dict: Dict[NodeType, Tensor],
    num_neighbors_dict: Dict[EdgeType, List[int]],
    node_time_dict: Optional[Dict[NodeType, Tensor]] = None,
    edge_time_dict: Optional[Dict[EdgeType, Tensor]] = None,
    seed_time_dict: Optional[Dict[NodeType, Tensor]] = None,
    edge_weight_dict: Optional[Dict[EdgeType, Tensor]] = None,
    csc: bool = False,
    replace: bool = False,
    directed: bool = True,
    disjoint: bool = False,
    temporal_strategy: str = 'uniform',
    return_edge_id: bool = True,
) -> Tuple[
        Dict[EdgeType, Tensor],
        Dict[EdgeType, Tensor],
        Dict[NodeType, Tensor],
        Optional[Dict[EdgeType, Tensor]],
        Dict[NodeType, List[int]],
        Dict[EdgeType, List[int]],
]:
    r"""Recursively samples neighbors from all node indices in :obj:`seed_dict`
    in the heterogeneous graph given by :obj:`(rowptr_dict, col_dict)` (same contract as
    :func:`pyg_lib.sampler.hetero_neighbor_sample`, pyg_lib/sampler/__init__.py:103-200).

    Relations are expanded in ``rowptr_dict`` order and seeds in ``seed_dict`` order, i.e. the
    reference's single-threaded order (its multi-threaded order races on the shared RNG).
    """
    # edge-type tuple <-> "a__b__c" key remapping exactly as the reference (:135-153, 183-191)
    src_node_types = {k[0] for k in rowptr_dict.keys()}
    dst_node_types = {k[-1] for k in rowptr_dict.keys()}
    node_types = sorted(src_node_types | dst_node_types | set(seed_dict.keys()))
    edge_types = list(rowptr_dict.keys())

    TO_REL_TYPE = {key: '__'.join(key) for key in edge_types}
    TO_EDGE_TYPE = {'__'.join(key): key for key in edge_types}

    rowptr_dict = {TO_REL_TYPE[k]: v for k, v in rowptr_dict.items()}
    col_dict = {TO_REL_TYPE[k]: v for k, v in col_dict.items()}
    num_neighbors_dict = {TO_REL_TYPE[k]: v for k, v in num_neighbors_dict.items()}
    if edge_time_dict is not None:
        edge_time_dict = {TO_REL_TYPE[k]: v for k, v in edge_time_dict.items()}
    if edge_weight_dict is not None:
        edge_weight_dict = {TO_REL_TYPE[k]: v for k, v in edge_weight_dict.items()}

    out = torch.ops.pyg.hetero_neighbor_sample(
        node_types,
        edge_types,
        rowptr_dict,
        col_dict,
        seed_dict,
        num_neighbors_dict,
        node_time_dict,
        edge_time_dict,
        seed_time_dict,
        edge_weight_dict,
        csc,
        replace,
        directed,
        disjoint,
        temporal_strategy,
        return_edge_id,
    )
    row_dict, col_dict, node_id_dict, edge_id_dict, num_nodes_per_hop_dict, num_edges_per_hop_dict = out

    row_dict = {TO_EDGE_TYPE[k]: v for k, v in row_dict.items()}
    col_dict = {TO_EDGE_TYPE[k]: v for k, v in col_dict.items()}
    if edge_id_dict is not None:
        edge_id_dict = {TO_EDGE_TYPE[k]: v for k, v in edge_id_dict.items()}
    num_edges_per_hop_dict = {TO_EDGE_TYPE[k]: v for k, v in num_edges_per_hop_dict.items()}

    return (
        row_dict,
        col_dict,
        node_id_dict,
        edge_id_dict,
        num_nodes_per_hop_dict,
        num_edges_per_hop_dict,
    )


__all__ = ['neighbor_sample', 'hetero_neighbor_sample']
