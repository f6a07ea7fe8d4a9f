"""Host-side mirror of ``pyg_lib.sampler`` for the hot path (pyg_lib/sampler/__init__.py:11-200).

Same names, arguments, defaults and return structure as the reference.  The graph (``rowptr``,
``col``) and the seeds live on a HIP device and the whole multi-hop expansion, including the
first-occurrence-ordered relabelling, runs there (the reference only has a single-threaded CPU
kernel).  Random numbers are drawn from PyTorch's global CPU generator exactly as the reference's
``RandintEngine`` does, so ``torch.manual_seed(s)`` yields bit-identical samples.
"""
import ctypes
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from pyg_lib_amd import _capi

NodeType = str
RelType = str
EdgeType = Tuple[str, str, str]

_I64_MIN, _I64_MAX = -2**63, 2**63 - 1


class _Relation(ctypes.Structure):
    """pyg_hip_relation"""
    _fields_ = [('rowptr', ctypes.c_void_p), ('num_rows', ctypes.c_int64), ('col', ctypes.c_void_p),
                ('num_cols', ctypes.c_int64), ('src_type', ctypes.c_int32), ('dst_type', ctypes.c_int32),
                ('num_neighbors_host', ctypes.c_void_p)]


class _SeedSet(ctypes.Structure):
    """pyg_hip_seed_set"""
    _fields_ = [('node_type', ctypes.c_int32), ('reserved', ctypes.c_int32), ('seed', ctypes.c_void_p),
                ('num_seed', ctypes.c_int64)]


_ALLOC = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
_FREE = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)
_RNG = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int)


class _Host(ctypes.Structure):
    """pyg_hip_sampler_host"""
    _fields_ = [('user', ctypes.c_void_p), ('alloc', _ALLOC), ('free', _FREE), ('rng_block', _RNG)]


class _Result(ctypes.Structure):
    """pyg_hip_sample_result"""
    _fields_ = [('node_id', ctypes.POINTER(ctypes.c_void_p)), ('num_nodes', ctypes.POINTER(ctypes.c_int64)),
                ('nodes_per_hop_host', ctypes.POINTER(ctypes.c_int64)),
                ('row', ctypes.POINTER(ctypes.c_void_p)), ('col', ctypes.POINTER(ctypes.c_void_p)),
                ('edge_id', ctypes.POINTER(ctypes.c_void_p)), ('num_edges', ctypes.POINTER(ctypes.c_int64)),
                ('edges_per_hop_host', ctypes.POINTER(ctypes.c_int64)), ('rng_blocks', ctypes.c_int64)]


_declared = False


def _lib():
    global _declared
    L = _capi.lib()
    if not _declared:
        L.pyg_hip_hetero_neighbor_sample.restype = ctypes.c_int
        L.pyg_hip_hetero_neighbor_sample.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _declared = True
    return L


class _HostServices:
    """Device memory from the PyTorch caching allocator; random words from the global CPU generator."""
    def __init__(self, device):
        self.device = device
        self.blocks = {}
        self.rng_buf = None
        self.error = None
        self.alloc_cb = _ALLOC(self._alloc)
        self.free_cb = _FREE(self._free)
        self.rng_cb = _RNG(self._rng)
        self.struct = _Host(None, self.alloc_cb, self.free_cb, self.rng_cb)

    def _alloc(self, _user, nbytes):
        try:
            n = (int(nbytes) + 7) // 8
            t = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
            self.blocks[t.data_ptr()] = t
            return t.data_ptr()
        except Exception as e:  # noqa: BLE001 - must not propagate through the C frame
            self.error = e
            return None

    def _free(self, _user, ptr):
        self.blocks.pop(ptr, None)

    def _rng(self, _user, words, first):
        # rand_engine.h:79-91: at::randint for the first prefetch, in-place random_ afterwards
        try:
            if first or self.rng_buf is None:
                self.rng_buf = torch.randint(_I64_MIN, _I64_MAX, (128,), dtype=torch.int64)
            else:
                self.rng_buf.random_(_I64_MIN, _I64_MAX)
            ctypes.memmove(words, self.rng_buf.data_ptr(), 128 * 8)
        except Exception as e:  # noqa: BLE001
            self.error = e

    def take(self, ptr, numel, shape=None):
        """Hand a result block over as a tensor (a view of the allocated block)."""
        t = self.blocks.pop(ptr)
        out = t[:numel]
        return out if shape is None else out.view(shape)


def _check_index_tensor(t: Tensor, name: str):
    if not t.is_contiguous():
        raise RuntimeError(f"Non-contiguous '{name}'")
    _capi.require_device(t, name)
    if t.dtype != torch.int64:
        raise RuntimeError(f"pyg_lib_amd: '{name}' must be int64 on the device path (got {t.dtype})")


def _sample(node_types: List[str], edge_types: List[EdgeType], rowptr_dict, col_dict, seed_dict,
            num_neighbors_dict, csc: bool, replace: bool, directed: bool, disjoint: bool, return_edge_id: bool):
    if not directed:
        # neighbor_kernel.cpp:501 / :815
        raise RuntimeError('Undirected subgraphs not yet supported')
    nt_index = {t: i for i, t in enumerate(node_types)}
    E = len(edge_types)
    Ls = {len(num_neighbors_dict[e]) for e in edge_types}
    L = max(Ls) if Ls else 0
    if len(Ls) > 1:
        raise RuntimeError('pyg_lib_amd: all relations must list the same number of hops')
    device = None
    for v in rowptr_dict.values():
        _check_index_tensor(v, 'rowptr')
        device = device or v.device
    for v in col_dict.values():
        _check_index_tensor(v, 'col')
    for v in seed_dict.values():
        _check_index_tensor(v, 'seed')
        device = device or v.device
    rels = (_Relation * max(E, 1))()
    keep = []
    for i, e in enumerate(edge_types):
        nn = (ctypes.c_int64 * max(L, 1))(*[int(x) for x in num_neighbors_dict[e]])
        keep.append(nn)
        rp, cl = rowptr_dict[e], col_dict[e]
        rels[i] = _Relation(rp.data_ptr(), rp.numel() - 1, cl.data_ptr(), cl.numel(), nt_index[e[0]],
                            nt_index[e[2]], ctypes.cast(nn, ctypes.c_void_p))
    seed_keys = list(seed_dict.keys())
    seeds = (_SeedSet * max(len(seed_keys), 1))()
    for i, k in enumerate(seed_keys):
        s = seed_dict[k]
        seeds[i] = _SeedSet(nt_index[k], 0, s.data_ptr(), s.numel())
    T = len(node_types)
    node_id = (ctypes.c_void_p * T)()
    num_nodes = (ctypes.c_int64 * T)()
    nodes_per_hop = (ctypes.c_int64 * (T * (L + 1)))()
    row = (ctypes.c_void_p * max(E, 1))()
    col = (ctypes.c_void_p * max(E, 1))()
    eid = (ctypes.c_void_p * max(E, 1))()
    num_edges = (ctypes.c_int64 * max(E, 1))()
    edges_per_hop = (ctypes.c_int64 * max(E * L, 1))()
    res = _Result(node_id, num_nodes, nodes_per_hop, row, col, eid, num_edges, edges_per_hop, 0)
    host = _HostServices(device)
    with torch.cuda.device(device):
        rc = _lib().pyg_hip_hetero_neighbor_sample(T, E, ctypes.byref(rels), len(seed_keys), ctypes.byref(seeds), L,
                                                   int(csc), int(replace), int(disjoint), int(return_edge_id),
                                                   ctypes.byref(host.struct), ctypes.byref(res),
                                                   _capi.stream_ptr(device))
    if host.error is not None:
        raise host.error
    _capi.check(rc)
    out_nodes, out_nhops = {}, {}
    for i, t in enumerate(node_types):
        n = int(num_nodes[i])
        out_nodes[t] = host.take(node_id[i], 2 * n, (n, 2)) if disjoint else host.take(node_id[i], n)
        out_nhops[t] = [int(nodes_per_hop[i * (L + 1) + l]) for l in range(L + 1)]
    out_row, out_col, out_eid, out_ehops = {}, {}, {}, {}
    for i, e in enumerate(edge_types):
        n = int(num_edges[i])
        out_row[e] = host.take(row[i], n)
        out_col[e] = host.take(col[i], n)
        if return_edge_id:
            out_eid[e] = host.take(eid[i], n)
        out_ehops[e] = [int(edges_per_hop[i * L + l]) for l in range(L)]
    return out_row, out_col, out_nodes, (out_eid if return_edge_id else None), out_nhops, out_ehops


def _unsupported(name):
    raise RuntimeError(f"pyg_lib_amd: '{name}' sampling is not implemented on the HIP device path yet "
                       f"(SURVEY.md 8(f) N4); refusing to fall back to a CPU kernel")


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

    Returns:
        Row indices, col indices of the returned subtree/subgraph, original node indices of all
        sampled nodes, optionally the indices of the sampled edges in the original graph, and the
        number of sampled nodes / edges per hop.
    """
    # precondition checks of the reference kernel, neighbor_kernel.cpp:354-380
    if (node_time is not None or edge_time is not None) and not disjoint:
        raise RuntimeError('Temporal sampling needs to create disjoint subgraphs')
    if node_time is not None and edge_time is not None:
        raise RuntimeError('Only one of node-level or edge-level sampling is supported ')
    if temporal_strategy not in ('uniform', 'last'):
        raise RuntimeError('No valid temporal strategy found')
    if node_time is not None or edge_time is not None:
        _unsupported('temporal')
    if edge_weight is not None:
        _unsupported('biased')
    et = ('_', '_', '_')
    out = _sample(['_'], [et], {et: rowptr}, {et: col}, {'_': seed}, {et: list(num_neighbors)}, csc, replace,
                  directed, disjoint, return_edge_id)
    row_d, col_d, node_d, eid_d, nh, eh = out
    return row_d[et], col_d[et], node_d['_'], (eid_d[et] if eid_d is not None else None), nh['_'], eh[et]


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
    if (node_time_dict is not None or edge_time_dict is not None) and not disjoint:
        raise RuntimeError('Node temporal sampling needs to create disjoint subgraphs')
    if temporal_strategy not in ('uniform', 'last'):
        raise RuntimeError('No valid temporal strategy found')
    if node_time_dict is not None or edge_time_dict is not None:
        _unsupported('temporal')
    if edge_weight_dict is not None:
        _unsupported('biased')
    # node / edge type discovery as in the reference (:135-138); sorted for a deterministic order
    src_node_types = {k[0] for k in rowptr_dict.keys()}
    dst_node_types = {k[-1] for k in rowptr_dict.keys()}
    node_types = sorted(src_node_types | dst_node_types | set(seed_dict.keys()))
    edge_types = list(rowptr_dict.keys())
    return _sample(node_types, edge_types, rowptr_dict, col_dict, seed_dict, num_neighbors_dict, csc, replace,
                   directed, disjoint, return_edge_id)


__all__ = ['neighbor_sample', 'hetero_neighbor_sample']
