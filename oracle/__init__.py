"""CPU oracle for the pyg-lib hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``pyg_lib_amd``) never does; it fails loudly without its HIP library.

The arithmetic lives in the C files next to this one (``oracle_*.c``, each citing the reference
``file:line`` it restates); this module is a thin ctypes/numpy wrapper around ``liboracle.so``.
"""
import ctypes
import os
import os.path as osp
import subprocess

import numpy as np

_HERE = osp.dirname(osp.abspath(__file__))
_LIB = None

F32, F64, F16, BF16 = 0, 1, 2, 3


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (a few seconds)."""
    so = osp.join(_HERE, 'liboracle.so')
    srcs = [osp.join(_HERE, f) for f in sorted(os.listdir(_HERE))
            if f.startswith('oracle_') and (f.endswith('.c') or f.endswith('.cpp'))]
    if force or not osp.exists(so) or any(osp.getmtime(s) > osp.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'liboracle.so'])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _declare(_LIB)
    return _LIB


_i64p = ctypes.POINTER(ctypes.c_int64)
_FILL = ctypes.CFUNCTYPE(None, ctypes.c_void_p, _i64p)


def _declare(L):
    L.oracle_segment_matmul.restype = ctypes.c_int
    L.oracle_segment_matmul.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int64]
    L.oracle_matmul.restype = ctypes.c_int
    L.oracle_matmul.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    L.oracle_mt19937_words.restype = None
    L.oracle_mt19937_words.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64]
    L.oracle_mt19937_word_after.restype = ctypes.c_int64
    L.oracle_mt19937_word_after.argtypes = [ctypes.c_uint64, ctypes.c_int64]
    L.oracle_biased_log_f32.restype = None
    L.oracle_biased_log_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.oracle_hetero_neighbor_sample.restype = ctypes.c_void_p
    L.oracle_hetero_neighbor_sample.argtypes = [
        ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,  # types, et_src, et_dst
        ctypes.c_void_p, ctypes.c_void_p,  # rowptr**, col**
        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,  # seeds
        ctypes.c_void_p, ctypes.c_int,  # num_neighbors, L
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,  # node_time**, edge_time**, seed_time**
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,  # csc, replace, disjoint, last
        ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    L.oracle_hetero_neighbor_sample_w.restype = ctypes.c_void_p
    L.oracle_hetero_neighbor_sample_w.argtypes = (
        L.oracle_hetero_neighbor_sample.argtypes[:15] + [ctypes.c_void_p, ctypes.c_void_p] +  # weight**, is_f64*
        L.oracle_hetero_neighbor_sample.argtypes[15:])
    L.oracle_topk_desc_f32.restype = None
    L.oracle_topk_desc_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    L.oracle_topk_desc_f64.restype = None
    L.oracle_topk_desc_f64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    L.oracle_sample_free.restype = None
    L.oracle_sample_free.argtypes = [ctypes.c_void_p]
    for name in ('oracle_sample_num_nodes', 'oracle_sample_num_edges'):
        getattr(L, name).restype = ctypes.c_int64
        getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_int]
    for name in ('oracle_sample_rng_blocks', 'oracle_sample_rng_draws', 'oracle_sample_rng_raw_draws'):
        getattr(L, name).restype = ctypes.c_int64
        getattr(L, name).argtypes = [ctypes.c_void_p]
    L.oracle_sample_copy_nodes.restype = None
    L.oracle_sample_copy_nodes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.oracle_sample_copy_edges.restype = None
    L.oracle_sample_copy_edges.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p]
    L.oracle_sample_copy_hops.restype = None
    L.oracle_sample_copy_hops.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


# ---- helpers ---------------------------------------------------------------------------------

def _np_dtype_code(a: np.ndarray) -> int:
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.float64:
        return F64
    if a.dtype == np.float16:
        return F16
    raise TypeError(f'unsupported dtype {a.dtype} (pass bf16 as uint16 with dtype=BF16)')


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit patterns (uint16)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    lsb = (u >> 16) & 1
    return ((u + 0x7fff + lsb) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# ---- matmul ----------------------------------------------------------------------------------

def segment_matmul(inputs: np.ndarray, ptr: np.ndarray, other: np.ndarray, bias=None, dtype=None) -> np.ndarray:
    """out[ptr[b]:ptr[b+1]] = inputs[ptr[b]:ptr[b+1]] @ other[b] (+ bias[b]).

    bf16 tensors are passed as uint16 bit patterns with ``dtype=BF16``.
    Rows outside [ptr[0], ptr[-1]) are returned as zeros.
    """
    code = dtype if dtype is not None else _np_dtype_code(inputs)
    inputs = np.ascontiguousarray(inputs)
    other = np.ascontiguousarray(other)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    N, K = inputs.shape
    B, K2, M = other.shape
    assert K == K2 and ptr.size == B + 1
    out = np.zeros((N, M), dtype=inputs.dtype)
    if bias is not None:
        bias = np.ascontiguousarray(bias)
    rc = lib().oracle_segment_matmul(code, _ptr(inputs), _ptr(ptr), _ptr(other), _ptr(bias), _ptr(out), N, K, M, B)
    if rc != 0:
        raise RuntimeError('oracle_segment_matmul: invalid ptr')
    return out


def matmul(a: np.ndarray, b: np.ndarray, dtype=None) -> np.ndarray:
    code = dtype if dtype is not None else _np_dtype_code(a)
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    out = np.zeros((a.shape[0], b.shape[1]), dtype=a.dtype)
    lib().oracle_matmul(code, _ptr(a), _ptr(b), _ptr(out), a.shape[0], a.shape[1], b.shape[1])
    return out


def grouped_matmul(inputs, others, dtype=None):
    return [matmul(a, b, dtype) for a, b in zip(inputs, others)]


# ---- RNG ---------------------------------------------------------------------------------------

def mt19937_words(seed: int, n: int) -> np.ndarray:
    """First n values of torch.randint(INT64_MIN, INT64_MAX, (n,)) after torch.manual_seed(seed)."""
    out = np.zeros(n, dtype=np.int64)
    lib().oracle_mt19937_words(seed & 0xFFFFFFFFFFFFFFFF, _ptr(out), n)
    return out


# ---- sampler -----------------------------------------------------------------------------------

def mt19937_word_after(seed: int, skip32: int) -> int:
    """torch.randint(INT64_MIN, INT64_MAX, (1,)) after manual_seed(seed) and `skip32` 32-bit outputs."""
    return int(lib().oracle_mt19937_word_after(seed & 0xFFFFFFFFFFFFFFFF, int(skip32)))


def biased_log_f32(u: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(u, dtype=np.float32)
    out = np.empty_like(u)
    lib().oracle_biased_log_f32(_ptr(u), _ptr(out), u.size)
    return out


def topk_desc(keys: np.ndarray, k: int) -> np.ndarray:
    """Indices of Tensor.topk(k) (largest, sorted) of a 1-D float32 / float64 array, libtorch's tie order."""
    keys = np.ascontiguousarray(keys)
    idx = np.empty(k, dtype=np.int64)
    fn = lib().oracle_topk_desc_f64 if keys.dtype == np.float64 else lib().oracle_topk_desc_f32
    assert keys.dtype in (np.float32, np.float64)
    fn(_ptr(keys), keys.size, int(k), _ptr(idx))
    return idx


def _pp(arrs):
    """array of pointers (NULL for None)"""
    t = (ctypes.c_void_p * max(len(arrs), 1))()
    for i, a in enumerate(arrs):
        t[i] = None if a is None else a.ctypes.data
    return t


def _c64(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a), dtype=np.int64)


def hetero_neighbor_sample(node_types, edge_types, rowptr_dict, col_dict, seed_dict, num_neighbors_dict,
                           node_time_dict=None, edge_time_dict=None, seed_time_dict=None, csc=False,
                           replace=False, disjoint=False, temporal_strategy='uniform', return_edge_id=True,
                           rng_seed=0, fill=None, edge_weight_dict=None):
    """Restates pyg::hetero_neighbor_sample (single-threaded order).

    `edge_weight_dict`: float32 / float64 per-edge weights of the relations to sample with bias
    (replace=False only).

    Dict keys: node types are strings, edge types are (src, rel, dst) tuples.  Returns
    (row_dict, col_dict, node_id_dict, edge_id_dict|None, num_nodes_per_hop_dict,
    num_edges_per_hop_dict, info) with numpy int64 arrays; `info` holds RNG consumption.
    `rng_seed` plays the role of torch.manual_seed(seed) right before the call; `fill(buf128)`
    optionally replaces the word source.
    """
    nt_index = {t: i for i, t in enumerate(node_types)}
    E = len(edge_types)
    et_src = np.array([nt_index[e[0]] for e in edge_types], dtype=np.int32)
    et_dst = np.array([nt_index[e[2]] for e in edge_types], dtype=np.int32)
    rowptrs = [_c64(rowptr_dict[e]) for e in edge_types]
    cols = [_c64(col_dict[e]) for e in edge_types]
    seed_keys = list(seed_dict.keys())
    seed_types = np.array([nt_index[k] for k in seed_keys], dtype=np.int32)
    seeds = [_c64(seed_dict[k]) for k in seed_keys]
    seed_len = np.array([s.size for s in seeds], dtype=np.int64)
    L = len(next(iter(num_neighbors_dict.values()))) if E else 0
    nn = np.array([list(num_neighbors_dict[e]) for e in edge_types], dtype=np.int64).reshape(E, L)
    ntimes = [_c64(node_time_dict.get(t)) if node_time_dict else None for t in node_types]
    etimes = [_c64(edge_time_dict.get(e)) if edge_time_dict else None for e in edge_types]
    stimes = [_c64(seed_time_dict.get(k)) if seed_time_dict else None for k in seed_keys]
    weights = [None] * E
    w64 = np.zeros(max(E, 1), dtype=np.int32)
    if edge_weight_dict:
        if node_time_dict or edge_time_dict:
            raise RuntimeError('Biased temporal sampling not yet supported')
        for i, e in enumerate(edge_types):
            w = edge_weight_dict.get(e)
            if w is not None:
                w = np.ascontiguousarray(w)
                assert w.dtype in (np.float32, np.float64)
                weights[i] = w
                w64[i] = int(w.dtype == np.float64)
    status = ctypes.c_int(0)
    cb = None
    if fill is not None:
        def _cb(_user, buf):
            arr = np.ctypeslib.as_array(buf, shape=(128,))
            fill(arr)
        cb = _FILL(_cb)
    L_ = lib()
    h = L_.oracle_hetero_neighbor_sample_w(
        len(node_types), E, _ptr(et_src), _ptr(et_dst), _pp(rowptrs), _pp(cols), len(seed_keys), _ptr(seed_types),
        _pp(seeds), _ptr(seed_len), _ptr(nn), L, _pp(ntimes), _pp(etimes), _pp(stimes), _pp(weights), _ptr(w64),
        int(csc), int(replace),
        int(disjoint), int(temporal_strategy == 'last'), rng_seed & 0xFFFFFFFFFFFFFFFF,
        ctypes.cast(cb, ctypes.c_void_p) if cb is not None else None, None, ctypes.byref(status))
    try:
        if status.value == -2:
            raise NotImplementedError('biased sampling with replacement and one draw per node (at::multinomial\'s '
                                      'exponential_ path) / with an external word source is not restated')
        if status.value == -3:
            raise RuntimeError('invalid multinomial distribution')
        if status.value != 0:
            raise RuntimeError('Found invalid non-sorted temporal neighborhood')
        rows, colsd, eids, nodes, nhops, ehops = {}, {}, {}, {}, {}, {}
        for i, t in enumerate(node_types):
            n = L_.oracle_sample_num_nodes(h, i)
            out = np.zeros((n, 2) if disjoint else (n,), dtype=np.int64)
            L_.oracle_sample_copy_nodes(h, i, _ptr(out))
            nodes[t] = out
            hops = np.zeros(L + 1, dtype=np.int64)
            L_.oracle_sample_copy_hops(h, i, 0, _ptr(hops))
            nhops[t] = hops.tolist()
        for i, e in enumerate(edge_types):
            n = L_.oracle_sample_num_edges(h, i)
            r = np.zeros(n, dtype=np.int64)
            c = np.zeros(n, dtype=np.int64)
            d = np.zeros(n, dtype=np.int64)
            L_.oracle_sample_copy_edges(h, i, _ptr(r), _ptr(c), _ptr(d))
            if csc:
                r, c = c, r  # get_sampled_edges(csc) swaps (neighbor_kernel.cpp:155-159)
            rows[e], colsd[e], eids[e] = r, c, d
            hops = np.zeros(L, dtype=np.int64)
            L_.oracle_sample_copy_hops(h, i, 1, _ptr(hops))
            ehops[e] = hops.tolist()
        info = {'rng_blocks': L_.oracle_sample_rng_blocks(h), 'rng_draws': L_.oracle_sample_rng_draws(h),
                'rng_raw_draws': L_.oracle_sample_rng_raw_draws(h)}
    finally:
        L_.oracle_sample_free(h)
    return rows, colsd, nodes, (eids if return_edge_id else None), nhops, ehops, info


def neighbor_sample(rowptr, col, seed, num_neighbors, node_time=None, edge_time=None, seed_time=None, csc=False,
                    replace=False, directed=True, disjoint=False, temporal_strategy='uniform',
                    return_edge_id=True, rng_seed=0, fill=None, edge_weight=None):
    """Restates pyg::neighbor_sample. Returns (row, col, node_id, edge_id|None, nodes_per_hop,
    edges_per_hop, info)."""
    if not directed:
        raise RuntimeError('Undirected subgraphs not yet supported')
    if (node_time is not None or edge_time is not None) and not disjoint:
        raise RuntimeError('Temporal sampling needs to create disjoint subgraphs')
    et = ('n', 'to', 'n')
    out = hetero_neighbor_sample(
        ['n'], [et], {et: rowptr}, {et: col}, {'n': seed}, {et: list(num_neighbors)},
        node_time_dict=None if node_time is None else {'n': node_time},
        edge_time_dict=None if edge_time is None else {et: edge_time},
        seed_time_dict=None if seed_time is None else {'n': seed_time},
        csc=csc, replace=replace, disjoint=disjoint, temporal_strategy=temporal_strategy,
        return_edge_id=return_edge_id, rng_seed=rng_seed, fill=fill,
        edge_weight_dict=None if edge_weight is None else {et: edge_weight})
    rows, cols, nodes, eids, nh, eh, info = out
    return rows[et], cols[et], nodes['n'], (eids[et] if eids is not None else None), nh['n'], eh[et], info


# ---- scatter / segment_coo / gather_coo / index_sort ---------------------------------------------
SUM, MUL, MIN, MAX = 0, 1, 2, 3
I8, U8, I16, I32, I64 = 4, 5, 6, 7, 8
_NP_CODES = {np.dtype(np.float32): F32, np.dtype(np.float64): F64, np.dtype(np.float16): F16,
             np.dtype(np.int8): I8, np.dtype(np.uint8): U8, np.dtype(np.int16): I16,
             np.dtype(np.int32): I32, np.dtype(np.int64): I64}


def _code(a, dtype):
    return dtype if dtype is not None else _NP_CODES[a.dtype]


def _declare_reduce(L):
    if getattr(L, '_reduce_declared', False):
        return
    c = ctypes
    L.oracle_scatter.restype = c.c_int
    L.oracle_scatter.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int,
                                 c.c_int64, c.c_int64, c.c_int64, c.c_int64]
    L.oracle_segment_sum_coo.restype = c.c_int
    L.oracle_segment_sum_coo.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_int64,
                                         c.c_int64, c.c_int64]
    L.oracle_gather_coo.restype = c.c_int
    L.oracle_gather_coo.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_int64, c.c_int64,
                                    c.c_int64]
    L.oracle_index_sort.restype = c.c_int
    L.oracle_index_sort.argtypes = [c.c_int, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
    L.oracle_segment_csr.restype = c.c_int
    L.oracle_segment_csr.argtypes = [c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int,
                                     c.c_int64, c.c_int64, c.c_int64, c.c_int64]
    L.oracle_gather_csr.restype = c.c_int
    L.oracle_gather_csr.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_int64, c.c_int64,
                                    c.c_int64]
    L.oracle_softmax_csr.restype = None
    L.oracle_softmax_csr.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_int64, c.c_int64, c.c_int64]
    L.oracle_softmax_csr_backward.restype = None
    L.oracle_softmax_csr_backward.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_int64,
                                              c.c_int64, c.c_int64]
    L._reduce_declared = True


def _identity(op, a, dtype):
    """numeric_limits<T>::max() / lowest() in the storage representation of `a`."""
    code = _code(a, dtype)
    if code == BF16:
        return np.uint16(0x7f7f if op == MIN else 0xff7f)
    if code == F16:
        return np.float16(65504.0 if op == MIN else -65504.0)
    if np.issubdtype(a.dtype, np.floating):
        return np.finfo(a.dtype).max if op == MIN else np.finfo(a.dtype).min
    return np.iinfo(a.dtype).max if op == MIN else np.iinfo(a.dtype).min


def _bcast_index(index, src, dim):
    """pyg_lib/csrc/ops/utils.h:22-34"""
    idx = np.asarray(index, dtype=np.int64)
    if idx.ndim == 1:
        idx = idx.reshape((1,) * dim + idx.shape)
    while idx.ndim < src.ndim:
        idx = idx[..., None]
    return np.ascontiguousarray(np.broadcast_to(idx, src.shape))


def scatter(op, src, index, dim=-1, out=None, dim_size=None, dtype=None):
    """scatter_{sum,mul,min,max}: returns (out, arg_out or None).  `out` (if given) is updated in place
    semantics-wise but a new array is returned."""
    L = lib()
    _declare_reduce(L)
    src = np.ascontiguousarray(src)
    dim = dim + src.ndim if dim < 0 else dim
    idx = _bcast_index(index, src, dim)
    fresh = out is None
    if fresh:
        n = dim_size if dim_size is not None else (0 if idx.size == 0 else int(idx.max()) + 1)
        shape = list(src.shape)
        shape[dim] = n
        if op == SUM:
            out = np.zeros(shape, dtype=src.dtype)
        elif op == MUL:
            out = np.ones(shape, dtype=src.dtype) if _code(src, dtype) != BF16 else np.full(shape, 0x3f80, np.uint16)
        else:
            out = np.full(shape, _identity(op, src, dtype), dtype=src.dtype)
    else:
        out = np.ascontiguousarray(out).copy()
    B = int(np.prod(src.shape[:dim], dtype=np.int64))
    E = src.shape[dim]
    K = int(np.prod(src.shape[dim + 1:], dtype=np.int64))
    N = out.shape[dim]
    arg = np.full(out.shape, E, dtype=np.int64) if op in (MIN, MAX) else None
    if src.size:
        rc = L.oracle_scatter(op, _code(src, dtype), _ptr(src), _ptr(idx), _ptr(out), _ptr(arg), int(fresh), B, E, K, N)
        if rc != 0:
            raise RuntimeError('oracle_scatter: index out of range')
    elif op in (MIN, MAX) and fresh:
        out[...] = 0
    return out, arg


def scatter_mean(src, index, dim=-1, out=None, dim_size=None, dtype=None):
    """ops/autograd/scatter_kernel.cpp:161-233 (floating dtypes only in this oracle wrapper)."""
    src = np.ascontiguousarray(src)
    d = dim + src.ndim if dim < 0 else dim
    s, _ = scatter(SUM, src, index, d, out, dim_size, dtype)
    idx = _bcast_index(index, src, d)
    ones = np.ones(src.shape, dtype=np.float64)
    cnt, _ = scatter(SUM, ones, idx, d, None, s.shape[d])
    cnt[cnt < 1] = 1
    if dtype == BF16:
        res = bf16_bits_to_f32(s) / cnt.astype(np.float32)
        return f32_to_bf16_bits(res)
    if np.issubdtype(src.dtype, np.floating):
        return (s / cnt.astype(s.dtype)).astype(s.dtype)
    return np.floor_divide(s, cnt.astype(s.dtype))


def _coo_shapes(src, index):
    index = np.asarray(index, dtype=np.int64)
    dim = index.ndim - 1
    idx = np.ascontiguousarray(np.broadcast_to(index, src.shape[:index.ndim]))
    B = int(np.prod(idx.shape[:dim], dtype=np.int64))
    E = src.shape[dim]
    K = int(np.prod(src.shape[index.ndim:], dtype=np.int64))
    return idx, dim, B, E, K


def segment_sum_coo(src, index, out=None, dim_size=None, dtype=None):
    L = lib()
    _declare_reduce(L)
    src = np.ascontiguousarray(src)
    idx, dim, B, E, K = _coo_shapes(src, index)
    if out is None:
        n = dim_size if dim_size is not None else (0 if idx.size == 0 else int(idx[..., -1].max()) + 1)
        shape = list(src.shape)
        shape[dim] = n
        out = np.zeros(shape, dtype=src.dtype)
    else:
        out = np.ascontiguousarray(out).copy()
    if src.size:
        rc = L.oracle_segment_sum_coo(_code(src, dtype), _ptr(src), _ptr(idx), _ptr(out), B, E, K, out.shape[dim])
        if rc != 0:
            raise RuntimeError('oracle_segment_sum_coo: index out of range')
    return out


def segment_minmax_coo(op, src, index, out=None, dim_size=None, dtype=None):
    """segment_{min,max}_coo == scatter_{min,max} with the [.., E] index broadcast over trailing dims."""
    src = np.ascontiguousarray(src)
    idx, dim, B, E, K = _coo_shapes(src, index)
    full = idx.reshape(idx.shape + (1,) * (src.ndim - idx.ndim))
    if out is None and dim_size is None:
        dim_size = 0 if idx.size == 0 else int(idx[..., -1].max()) + 1
    return scatter(op, src, np.broadcast_to(full, src.shape), dim, out, dim_size, dtype)


def gather_coo(src, index, dtype=None):
    L = lib()
    _declare_reduce(L)
    src = np.ascontiguousarray(src)
    index = np.ascontiguousarray(index, dtype=np.int64)
    dim = index.ndim - 1
    B = int(np.prod(index.shape[:dim], dtype=np.int64))
    E = index.shape[dim]
    N = src.shape[dim]
    K = int(np.prod(src.shape[index.ndim:], dtype=np.int64))
    shape = list(src.shape)
    shape[dim] = E
    out = np.zeros(shape, dtype=src.dtype)
    if out.size and src.size:
        rc = L.oracle_gather_coo(_code(src, dtype), _ptr(src), _ptr(index), _ptr(out), B, E, K, N)
        if rc != 0:
            raise RuntimeError('oracle_gather_coo: index out of range')
    return out


def index_sort(keys):
    L = lib()
    _declare_reduce(L)
    keys = np.ascontiguousarray(keys)
    if keys.ndim != 1:
        raise RuntimeError('Input should be 1-dimensional.')
    out = np.zeros_like(keys)
    idx = np.zeros(keys.size, dtype=np.int64)
    rc = L.oracle_index_sort(_NP_CODES[keys.dtype], _ptr(keys), keys.size, _ptr(out), _ptr(idx))
    if rc != 0:
        raise RuntimeError('Input should contain integral values.')
    return out, idx


def dist_neighbor_sample(rowptr, col, seed, num_neighbors, node_time=None, edge_time=None, seed_time=None, csc=False,
                         replace=False, directed=True, disjoint=False, temporal_strategy='uniform', rng_seed=0,
                         edge_weight=None):
    """Restates pyg::dist_neighbor_sample. Returns (node_id, edge_id, cumsum_neighbors_per_node, info)."""
    if (node_time is not None or edge_time is not None) and not disjoint:
        raise RuntimeError('Temporal sampling needs to create disjoint subgraphs')
    L = lib()
    c = ctypes
    L.oracle_dist_neighbor_sample_w.restype = c.c_int64
    L.oracle_dist_neighbor_sample_w.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_int64, c.c_void_p,
                                                c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int,
                                                c.c_uint64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    rowptr, col, seed = _c64(rowptr), _c64(col), _c64(seed)
    nt, et, st = _c64(node_time), _c64(edge_time), _c64(seed_time)
    w = None if edge_weight is None else np.ascontiguousarray(edge_weight)
    assert w is None or w.dtype in (np.float32, np.float64)
    S = seed.size
    args = (_ptr(rowptr), _ptr(col), _ptr(seed), S, int(num_neighbors), _ptr(nt), _ptr(et), _ptr(st), _ptr(w),
            int(w is not None and w.dtype == np.float64), int(replace), int(disjoint), int(temporal_strategy == 'last'),
            rng_seed & 0xFFFFFFFFFFFFFFFF)
    E = L.oracle_dist_neighbor_sample_w(*args, None, None, None, None, None)
    if E == -2:
        raise NotImplementedError('biased sampling with replacement and one draw per node is not restated')
    if E == -3:
        raise RuntimeError('invalid multinomial distribution')
    if E < 0:
        raise RuntimeError('Found invalid non-sorted temporal neighborhood')
    nodes = np.zeros((S + E, 2) if disjoint else (S + E,), dtype=np.int64)
    edges = np.zeros(E, dtype=np.int64)
    cumsum = np.zeros(S + 1, dtype=np.int64)
    blocks = ctypes.c_int64(0)
    raw = ctypes.c_int64(0)
    L.oracle_dist_neighbor_sample_w(*args, _ptr(nodes), _ptr(edges), _ptr(cumsum), ctypes.byref(blocks), ctypes.byref(raw))
    return nodes, edges, cumsum.tolist(), {'rng_blocks': blocks.value, 'rng_raw_draws': raw.value}


# ---- CSR family (segment_*_csr, gather_csr, softmax_csr) ----------------------------------------------
CSR_SUM, CSR_MEAN, CSR_MIN, CSR_MAX = 0, 1, 2, 3


def _csr_layout(src, indptr):
    """(indptr broadcast to [leading, rows + 1], leading, rows, dim) as segment_csr_kernel.cpp:44-58."""
    indptr = np.asarray(indptr, dtype=np.int64)
    if src.ndim < indptr.ndim:
        raise RuntimeError('src.dim() must be >= indptr.dim()')
    dim = indptr.ndim - 1
    shape = list(src.shape[:dim]) + [indptr.shape[-1]]
    ib = np.ascontiguousarray(np.broadcast_to(indptr, shape))
    leading = int(np.prod(shape[:-1], dtype=np.int64))
    return ib, leading, indptr.shape[-1] - 1, dim


def segment_csr(op, src, indptr, out=None, dtype=None):
    """segment_{sum,mean,min,max}_csr: returns (out, arg_out or None)."""
    L = lib()
    _declare_reduce(L)
    src = np.ascontiguousarray(src)
    ib, leading, rows, dim = _csr_layout(src, indptr)
    fresh = out is None
    shape = list(src.shape)
    shape[dim] = max(rows, 0)
    red = MIN if op == CSR_MIN else MAX
    if fresh or op == CSR_MEAN:
        if op in (CSR_SUM, CSR_MEAN):
            out = np.zeros(shape, dtype=src.dtype)
        else:
            out = np.full(shape, _identity(red, src, dtype), dtype=src.dtype)
    else:
        out = np.ascontiguousarray(out).copy()
    E = src.shape[dim]
    K = int(np.prod(src.shape[dim + 1:], dtype=np.int64))
    arg = np.full(out.shape, E, dtype=np.int64) if op in (CSR_MIN, CSR_MAX) else None
    if src.size:
        rc = L.oracle_segment_csr(op, _code(src, dtype), _ptr(src), _ptr(ib), _ptr(out), _ptr(arg), int(fresh),
                                  leading, rows, E, K)
        if rc != 0:
            raise RuntimeError(f'oracle_segment_csr failed ({rc})')
    elif op in (CSR_MIN, CSR_MAX) and fresh:
        out[...] = 0
    return out, arg


def gather_csr(src, indptr, out, dtype=None):
    """gather_csr into a copy of `out` (positions outside every row keep `out`'s contents)."""
    L = lib()
    _declare_reduce(L)
    src = np.ascontiguousarray(src)
    ib, leading, rows, dim = _csr_layout(src, indptr)
    out = np.ascontiguousarray(out).copy()
    E = out.shape[dim]
    K = int(np.prod(src.shape[dim + 1:], dtype=np.int64))
    if src.size:
        rc = L.oracle_gather_csr(_code(src, dtype), _ptr(src), _ptr(ib), _ptr(out), leading, rows, E, K)
        if rc != 0:
            raise RuntimeError('oracle_gather_csr: indptr out of range')
    return out


def _softmax_layout(src, dim):
    dim = dim + src.ndim if dim < 0 else dim
    outer = int(np.prod(src.shape[:dim], dtype=np.int64))
    inner = int(np.prod(src.shape[dim + 1:], dtype=np.int64))
    return outer, src.shape[dim], inner


def softmax_csr(src, ptr, dim=0):
    L = lib()
    _declare_reduce(L)
    src = np.ascontiguousarray(src, dtype=np.float32)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    outer, D, inner = _softmax_layout(src, dim)
    out = np.empty_like(src)
    L.oracle_softmax_csr(_ptr(src), _ptr(ptr), _ptr(out), outer, D, inner, ptr.size - 1)
    return out


def softmax_csr_backward(out, out_grad, ptr, dim=0):
    L = lib()
    _declare_reduce(L)
    out = np.ascontiguousarray(out, dtype=np.float32)
    out_grad = np.ascontiguousarray(out_grad, dtype=np.float32)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    outer, D, inner = _softmax_layout(out, dim)
    gin = np.empty_like(out)
    L.oracle_softmax_csr_backward(_ptr(out), _ptr(out_grad), _ptr(ptr), _ptr(gin), outer, D, inner, ptr.size - 1)
    return gin


# ---- distributed-sampling helpers (dist_relabel / dist_merge_outputs, homogeneous forms) ----------------
def relabel_neighborhood(seed, sampled_nodes_with_duplicates, num_sampled_neighbors_per_node, num_nodes=None,
                         batch=None, csc=False, disjoint=False):
    """pyg::relabel_neighborhood (pyg_lib/csrc/sampler/cpu/dist_relabel_kernel.cpp:30-94): Mapper ids in
    insertion order -- seeds first (`fill`: id = position of the first occurrence among distinct seeds;
    disjoint: key (i, seed[i])), then the sampled nodes in sequence; row = index of the source node."""
    seed = np.asarray(seed, dtype=np.int64)
    nodes = np.asarray(sampled_nodes_with_duplicates, dtype=np.int64)
    ids = {}
    for i, v in enumerate(seed.tolist()):
        ids.setdefault((i, v) if disjoint else v, len(ids))
    rows, cols = [], []
    j = 0
    for i, c in enumerate(num_sampled_neighbors_per_node):
        for _ in range(int(c)):
            key = (int(batch[j]), int(nodes[j])) if disjoint else int(nodes[j])
            cols.append(ids.setdefault(key, len(ids)))
            rows.append(i)
            j += 1
    row, col = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
    return (col, row) if csc else (row, col)


def hetero_relabel_neighborhood(node_types, edge_types, seed_dict, sampled_nodes_with_duplicates_dict,
                                num_sampled_neighbors_per_node_dict, num_nodes_dict=None, batch_dict=None,
                                csc=False, disjoint=False):
    """pyg::hetero_relabel_neighborhood (pyg_lib/csrc/sampler/cpu/dist_relabel_kernel.cpp:96-262, the
    single-threaded order): one Mapper per node type (seeds first; disjoint: batch ids run on across the seed
    types, :181-192); per layer, per edge type (in `edge_types` order), per source node i of that layer's range
    (:199-236) the next `count` nodes of the DESTINATION type's list are inserted; row = i.  The source ranges
    of the next layer start behind the largest range end of the edge types sharing the source type (:239-255).
    `num_sampled_neighbors_per_node_dict[edge_type]` is a list (layers) of lists (counts per source node)."""
    ids = {t: {} for t in node_types}
    b = 0
    for t, seed in seed_dict.items():
        for v in np.asarray(seed, dtype=np.int64).tolist():
            ids[t].setdefault((b, v) if disjoint else v, len(ids[t]))
            b += 1
    nodes = {t: np.asarray(sampled_nodes_with_duplicates_dict[t], dtype=np.int64) for t in node_types}
    pos = {t: 0 for t in node_types}
    rows = {k: [] for k in edge_types}
    cols = {k: [] for k in edge_types}
    counts = num_sampled_neighbors_per_node_dict
    slices = {k: (0, len(counts[k][0])) for k in edge_types}
    src_off = {t: 0 for t in node_types}
    num_layers = len(counts[edge_types[0]])
    for ell in range(num_layers):
        for k in edge_types:
            dst = k[2] if not csc else k[0]
            lo, hi = slices[k]
            for i in range(lo, hi):
                for _ in range(int(counts[k][ell][i - lo])):
                    j = pos[dst]
                    key = (int(batch_dict[dst][j]), int(nodes[dst][j])) if disjoint else int(nodes[dst][j])
                    cols[k].append(ids[dst].setdefault(key, len(ids[dst])))
                    rows[k].append(i)
                    pos[dst] = j + 1
        if ell < num_layers - 1:
            for k in edge_types:
                src = k[0] if not csc else k[2]
                src_off[src] = max(src_off[src], slices[k][1])
            for k in edge_types:
                src = k[0] if not csc else k[2]
                slices[k] = (src_off[src], src_off[src] + len(counts[k][ell + 1]))
    out_row, out_col = {}, {}
    for k in edge_types:
        r, c = np.asarray(rows[k], dtype=np.int64), np.asarray(cols[k], dtype=np.int64)
        out_row[k], out_col[k] = (c, r) if csc else (r, c)
    return out_row, out_col


def merge_sampler_outputs(node_ids, edge_ids, cumsum_neighbors_per_node, partition_ids, partition_orders,
                          num_partitions, num_neighbors, batch=None, disjoint=False):
    """pyg::merge_sampler_outputs (pyg_lib/csrc/sampler/cpu/dist_merge_outputs_kernel.cpp:17-138): the
    sampled neighbours of node j live in partition partition_ids[j] at order partition_orders[j]; outputs are
    the per-node segments concatenated in j order."""
    out_n, out_e, out_b, counts = [], [], [], []
    for j, (p, o) in enumerate(zip(partition_ids, partition_orders)):
        cs = cumsum_neighbors_per_node[p]
        bn, en = cs[o], cs[o + 1]
        out_n.append(np.asarray(node_ids[p], dtype=np.int64)[bn:en])
        out_e.append(np.asarray(edge_ids[p], dtype=np.int64)[bn - cs[0]:en - cs[0]])
        if disjoint:
            out_b.append(np.full(en - bn, int(batch[j]), dtype=np.int64))
        counts.append(en - bn)
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int64)
    return cat(out_n), cat(out_e), (cat(out_b) if disjoint else None), counts
