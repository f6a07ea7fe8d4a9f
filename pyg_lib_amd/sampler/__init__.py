"""``pyg_lib.sampler`` surface of the hot path: ``neighbor_sample`` and ``hetero_neighbor_sample``.

Argument names, order, defaults and the shape of the results follow the reference's Python front
(pyg_lib/sampler/__init__.py:11-200) so that callers (PyG's ``NeighborSampler``) need no change; everything
behind them is this repository's: the operators ``torch.ops.pyg.neighbor_sample`` /
``hetero_neighbor_sample`` are registered by ``libpyg.so`` and run the whole multi-hop expansion -- sampling,
first-occurrence relabelling, per-hop bookkeeping -- on the HIP device that holds the graph.  The random
stream is the global CPU generator's (continued on the device), so ``torch.manual_seed(s)`` reproduces the
reference's samples bit for bit.  CPU tensors take the ``CPU`` dispatch key's restatement of the same drivers
(csrc/binding/pyg_binding_cpu.cpp: host logic for tests and tooling, not a performance path).
"""
from typing import Dict, List, Optional, Tuple

import weakref

import torch
from torch import Tensor

NodeType = str
RelType = str
EdgeType = Tuple[str, str, str]

HomoOut = Tuple[Tensor, Tensor, Tensor, Optional[Tensor], List[int], List[int]]
HeteroOut = Tuple[Dict[EdgeType, Tensor], Dict[EdgeType, Tensor], Dict[NodeType, Tensor],
                  Optional[Dict[EdgeType, Tensor]], Dict[NodeType, List[int]], Dict[EdgeType, List[int]]]


# ---- which tensors are expanded-node outputs of these samplers -------------------------------------------------------
# The samplers emit a relation's edges grouped by the node they were sampled FOR, so the vector of expanded nodes -- `row`
# with csc=False, `col` with csc=True (pyg_lib/csrc/sampler/cpu/neighbor_kernel.cpp:147-159) -- is nondecreasing.
# pyg_lib_amd.rgcn's fused layer has an atomic-free kernel for exactly that (grouped=True); so that the usual pipeline
# sampler -> layer gets it without a flag, the wrappers below remember those tensors as they hand them out: by identity,
# weakly, together with the tensor's version counter -- a copy, a slice or a tensor moved to another device is a new
# tensor and not remembered, and one written to in place afterwards (`sort_`, `copy_`, `t[mask] = ...`) no longer counts
# (the kernel verifies the order on the device in any case).
_grouped_rows: Dict[int, Tuple["weakref.ref", int]] = {}


def _mark_grouped(t: Optional[Tensor]) -> None:
    if t is not None:
        key = id(t)
        _grouped_rows[key] = (weakref.ref(t, lambda _r, k=key: _grouped_rows.pop(k, None)), t._version)


def rows_are_grouped(t: Tensor) -> bool:
    r"""True if `t` is (the very tensor object of) an expanded-node output of one of this module's samplers -- ``row`` of a
    ``csc=False`` call, ``col`` of a ``csc=True`` call: nondecreasing by construction -- and has not been modified in
    place since it was returned."""
    e = _grouped_rows.get(id(t))
    return e is not None and e[0]() is t and e[1] == t._version


def neighbor_sample(rowptr: Tensor, col: Tensor, seed: Tensor, num_neighbors: List[int],
                    node_time: Optional[Tensor] = None, edge_time: Optional[Tensor] = None,
                    seed_time: Optional[Tensor] = None, edge_weight: Optional[Tensor] = None,
                    csc: bool = False, replace: bool = False, directed: bool = True, disjoint: bool = False,
                    temporal_strategy: str = 'uniform', return_edge_id: bool = True) -> HomoOut:
    """Multi-hop neighbour sampling on a homogeneous CSR graph that lives on a HIP device.

    ``rowptr`` / ``col`` are the CSR arrays (int64 or int32; with ``csc=True`` they are read as
    ``(colptr, row)`` and the returned row / col swap roles), ``seed`` the start nodes.  Hop ``l`` expands
    every node found in hop ``l - 1`` by ``num_neighbors[l]`` of its neighbours (``-1``: all of them),
    without replacement unless ``replace``.  ``disjoint`` keeps one subgraph per seed (node ids become
    ``(seed index, node)`` pairs).  Temporal sampling needs ``disjoint``: ``node_time`` (per node) or
    ``edge_time`` (per edge, with ``seed_time``) restrict a seed's subgraph to what existed at its time;
    ``temporal_strategy`` is ``'uniform'`` or ``'last'``.  ``edge_weight`` (float32 / float64, one per edge)
    turns the draw into weighted sampling without replacement.  ``directed=False`` is rejected, as in the
    reference.

    Returns ``(row, col, node_id, edge_id or None, num_nodes_per_hop, num_edges_per_hop)``: the sampled
    edges in local numbering (nodes numbered in order of first appearance, seeds first), the global id of
    every local node, the global edge ids unless ``return_edge_id=False``, and the per-hop counts.
    """
    args = (rowptr, col, seed, num_neighbors, node_time, edge_time, seed_time, edge_weight, csc, replace,
            directed, disjoint, temporal_strategy, return_edge_id)
    out = torch.ops.pyg.neighbor_sample(*args)
    _mark_grouped(out[1] if csc else out[0])
    return out


def _rel(edge_type: EdgeType) -> RelType:
    # the operator schema keys relations by one string (pyg_lib/csrc/utils/types.h: "src__rel__dst")
    return '__'.join(edge_type)


def _to_rel_keys(d):
    return None if d is None else {_rel(k): v for k, v in d.items()}


def hetero_neighbor_sample(rowptr_dict: Dict[EdgeType, Tensor], col_dict: Dict[EdgeType, Tensor],
                           seed_dict: Dict[NodeType, Tensor], num_neighbors_dict: Dict[EdgeType, List[int]],
                           node_time_dict: Optional[Dict[NodeType, Tensor]] = None,
                           edge_time_dict: Optional[Dict[EdgeType, Tensor]] = None,
                           seed_time_dict: Optional[Dict[NodeType, Tensor]] = None,
                           edge_weight_dict: Optional[Dict[EdgeType, Tensor]] = None, csc: bool = False,
                           replace: bool = False, directed: bool = True, disjoint: bool = False,
                           temporal_strategy: str = 'uniform', return_edge_id: bool = True) -> HeteroOut:
    """Heterogeneous counterpart of :func:`neighbor_sample`: one CSR per ``(src, rel, dst)`` edge type, seeds
    and times per node type, fan-outs (and optional weights / edge times) per edge type.

    Relations are expanded in the iteration order of ``rowptr_dict`` and seeds in that of ``seed_dict`` --
    the order of the reference's single-threaded kernel, which is the one its results are defined by (its
    multi-threaded variant races on the shared generator).  Results come back keyed by the caller's
    edge-type tuples / node types: ``(row_dict, col_dict, node_id_dict, edge_id_dict or None,
    num_nodes_per_hop_dict, num_edges_per_hop_dict)``.
    """
    edge_types = list(rowptr_dict)
    node_types = sorted({t for e in edge_types for t in (e[0], e[-1])} | set(seed_dict))
    back = {_rel(e): e for e in edge_types}

    out = torch.ops.pyg.hetero_neighbor_sample(
        node_types, edge_types, _to_rel_keys(rowptr_dict), _to_rel_keys(col_dict), seed_dict,
        _to_rel_keys(num_neighbors_dict), node_time_dict, _to_rel_keys(edge_time_dict), seed_time_dict,
        _to_rel_keys(edge_weight_dict), csc, replace, directed, disjoint, temporal_strategy, return_edge_id)
    rows, cols, node_ids, edge_ids, nodes_per_hop, edges_per_hop = out
    for t in (cols if csc else rows).values():
        _mark_grouped(t)

    def to_edge_keys(d):
        return None if d is None else {back[k]: v for k, v in d.items()}

    return (to_edge_keys(rows), to_edge_keys(cols), node_ids, to_edge_keys(edge_ids), nodes_per_hop,
            to_edge_keys(edges_per_hop))


def neighbor_sample_batched(rowptr: Tensor, col: Tensor, seeds: List[Tensor], num_neighbors: List[int],
                            generator_seeds: List[int], node_time: Optional[Tensor] = None,
                            edge_time: Optional[Tensor] = None, seed_times: Optional[List[Tensor]] = None,
                            edge_weight: Optional[Tensor] = None, csc: bool = False, replace: bool = False,
                            directed: bool = True, disjoint: bool = False, temporal_strategy: str = 'uniform',
                            return_edge_id: bool = True) -> List[HomoOut]:
    """``K = len(seeds)`` independent mini-batches in ONE call (this build only; the reference samples its epoch one batch
    at a time, benchmark/sampler/neighbor.py:101-121).

    Batch ``b`` is exactly ``torch.manual_seed(generator_seeds[b]); neighbor_sample(rowptr, col, seeds[b], ...)`` -- every
    output bit for bit -- but the batches are driven by a pool of host threads on private streams, so their chains of small
    dependent launches overlap on the device (a single batch cannot fill 256 CUs).  The process's default generator is
    not touched.  Returns the list of the ``K`` usual 6-tuples.  ``PYG_HIP_SAMPLER_LANES`` (default 8; 2 for heterogeneous
    graphs) caps the number of batches in flight.

    Stream contract: the outputs are allocated on the lanes' private streams and are complete when the call returns (the
    lanes are synchronised).  Consume them on the stream that was current when the call was made (or synchronise before
    dropping them): a later batched call orders its lanes behind the THEN-current stream before it reuses their blocks, so
    outputs read on another stream and dropped while that read is still queued could be handed out again under it.
    """
    rows, cols, nodes, eids, nph, eph = torch.ops.pyg.neighbor_sample_batched(
        rowptr, col, seeds, num_neighbors, generator_seeds, node_time, edge_time, seed_times, edge_weight, csc, replace,
        directed, disjoint, temporal_strategy, return_edge_id)
    nph, eph = nph.tolist(), eph.tolist()
    for t in (cols if csc else rows):
        _mark_grouped(t)
    return [(rows[b], cols[b], nodes[b], eids[b] if return_edge_id else None, nph[b], eph[b]) for b in range(len(seeds))]


def hetero_neighbor_sample_batched(rowptr_dict: Dict[EdgeType, Tensor], col_dict: Dict[EdgeType, Tensor],
                                   seed_dicts: List[Dict[NodeType, Tensor]],
                                   num_neighbors_dict: Dict[EdgeType, List[int]], generator_seeds: List[int],
                                   node_time_dict: Optional[Dict[NodeType, Tensor]] = None,
                                   edge_time_dict: Optional[Dict[EdgeType, Tensor]] = None,
                                   seed_time_dicts: Optional[List[Dict[NodeType, Tensor]]] = None,
                                   edge_weight_dict: Optional[Dict[EdgeType, Tensor]] = None,
                                   csc: bool = False, replace: bool = False, directed: bool = True, disjoint: bool = False,
                                   temporal_strategy: str = 'uniform', return_edge_id: bool = True) -> List[HeteroOut]:
    """Heterogeneous counterpart of :func:`neighbor_sample_batched`, with every mode of :func:`hetero_neighbor_sample`
    (node- / edge-level temporal sampling with one ``seed_time`` dict per batch, biased sampling; the reference has one entry
    for all of them, sampler/neighbor.cpp:137-147).  Batch ``b`` equals ``torch.manual_seed(generator_seeds[b]);
    hetero_neighbor_sample(rowptr_dict, col_dict, seed_dicts[b], num_neighbors_dict, ..., seed_time_dicts[b], ...)`` bit for bit."""
    edge_types = list(rowptr_dict)
    node_types = sorted({t for e in edge_types for t in (e[0], e[-1])} | {t for d in seed_dicts for t in d})
    back = {_rel(e): e for e in edge_types}
    rows, cols, nodes, eids, nph, eph = torch.ops.pyg.hetero_neighbor_sample_batched(
        node_types, edge_types, _to_rel_keys(rowptr_dict), _to_rel_keys(col_dict), seed_dicts,
        _to_rel_keys(num_neighbors_dict), generator_seeds, node_time_dict, _to_rel_keys(edge_time_dict), seed_time_dicts,
        _to_rel_keys(edge_weight_dict), csc, replace, directed, disjoint, temporal_strategy, return_edge_id)

    for d in (cols if csc else rows):
        for t in d.values():
            _mark_grouped(t)

    def to_edge_keys(d):
        return {back[k]: v for k, v in d.items()}

    return [(to_edge_keys(rows[b]), to_edge_keys(cols[b]), nodes[b], to_edge_keys(eids[b]) if return_edge_id else None,
             nph[b], to_edge_keys(eph[b])) for b in range(len(seed_dicts))]


def release_table_cache() -> int:
    """Hands the node tables the sampler keeps between calls (up to 8 x 128 MiB per device) back to the caching allocator;
    returns how many are still in use by running calls.  ``torch.cuda.empty_cache()`` afterwards returns the memory to
    the driver."""
    return int(torch.ops.pyg.sampler_release_table_cache())


def rng_carry_stats() -> Tuple[int, int]:
    """``(adopted, cold)``: sampler calls of this process that continued the random-word stream the previous call left on
    the device (same generator, untouched in between: no generation launch in front of any hop) / that looked for one and
    started cold (``pyg_hip_sampler_rng_carry_stats`` in include/pyg_hip.h)."""
    import ctypes
    from .. import _capi
    a, k = ctypes.c_int64(0), ctypes.c_int64(0)
    _capi.check(_capi.lib().pyg_hip_sampler_rng_carry_stats(ctypes.byref(a), ctypes.byref(k)))
    return int(a.value), int(k.value)


def last_mode() -> str:
    """Driver of the calling thread's last sampler call: 'fused', 'queued' or 'synchronising' (diagnostics; see
    ``pyg_hip_sampler_last_mode`` in include/pyg_hip.h)."""
    import ctypes
    from .. import _capi
    L = _capi.lib()
    L.pyg_hip_sampler_last_mode.restype = ctypes.c_char_p
    return L.pyg_hip_sampler_last_mode().decode()


__all__ = ['neighbor_sample', 'hetero_neighbor_sample', 'neighbor_sample_batched', 'hetero_neighbor_sample_batched',
           'release_table_cache', 'last_mode', 'rng_carry_stats']
