"""Host-side mirror of ``pyg_lib.ops`` for the hot path (pyg_lib/ops/__init__.py).

Same names, argument meaning, defaults and error behaviour as the reference; every op runs a
hand-written gfx950 kernel through the C-ABI of include/pyg_hip.h.  Tensors must live on a HIP
device -- there is no CPU kernel and no Triton path in this package.
"""
from typing import List, Optional, Tuple

import contextlib

import torch
from torch import Tensor

from pyg_lib_amd import _capi


def segment_matmul(
    inputs: Tensor,
    ptr: Tensor,
    other: Tensor,
    bias: Optional[Tensor] = None,
) -> Tensor:
    """``out[ptr[b]:ptr[b + 1]] = inputs[ptr[b]:ptr[b + 1]] @ other[b] (+ bias[b])`` for every segment ``b`` --
    one persistent MFMA launch for all segments (interface of the reference's
    ``pyg_lib.ops.segment_matmul``, pyg_lib/ops/__init__.py:137-172).

    ``inputs`` is ``[N, K]`` on a HIP device, ``ptr`` the ``B + 1`` row boundaries (host or device; neither
    placement synchronises), ``other`` ``[B, K, M]``, ``bias`` optionally ``[B, M]``.  Returns ``[N, M]``.
    Differentiable in ``inputs``, ``other`` and ``bias``.
    """
    needs_grad = torch.is_grad_enabled() and (inputs.requires_grad or other.requires_grad or
                                              (bias is not None and bias.requires_grad))
    if bias is not None and not needs_grad:
        # bias as a fused GEMM epilogue (the reference: B python-side slice adds, :169-171)
        return torch.ops.pyg.segment_matmul_bias(inputs, ptr, other, bias)
    out = torch.ops.pyg.segment_matmul(inputs, ptr, other)
    if bias is not None:
        for i in range(ptr.numel() - 1):
            out[ptr[i]:ptr[i + 1]] += bias[i]
    return out


# ---------------------------------------------------------------------------------------------------
# grouped_matmul
# ---------------------------------------------------------------------------------------------------

def _grouped_matmul_fwd(inputs: List[Tensor], others: List[Tensor]) -> List[Tensor]:
    return list(torch.ops.pyg.grouped_matmul(list(inputs), list(others)))


class _GroupedMatmul(torch.autograd.Function):
    """Mirrors GroupedMatmul (pyg_lib/ops/__init__.py:59-96) without the pytree indirection:
    the flat argument tuple is `inputs + others`."""
    @staticmethod
    def forward(ctx, *args: Tensor):
        ctx.save_for_backward(*args)
        n = len(args) // 2
        outs = _grouped_matmul_fwd(list(args[:n]), list(args[n:]))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *outs_grad: Tensor):
        args = ctx.saved_tensors
        n = len(outs_grad)
        inputs, others = list(args[:n]), list(args[n:])
        inputs_grad = [None] * n
        if any(x.requires_grad for x in inputs):
            inputs_grad = _grouped_matmul_fwd([g.contiguous() for g in outs_grad], [o.t() for o in others])
        others_grad = [None] * n
        if any(o.requires_grad for o in others):
            others_grad = _grouped_matmul_fwd([x.t() for x in inputs], [g.contiguous() for g in outs_grad])
        return tuple(inputs_grad + others_grad)


def grouped_matmul(
    inputs: List[Tensor],
    others: List[Tensor],
    biases: Optional[List[Tensor]] = None,
) -> List[Tensor]:
    """``outs[i] = inputs[i] @ others[i] (+ biases[i])`` for lists of independent 2-D operands
    (``[N_i, K_i]`` x ``[K_i, M_i]``) in one launch (interface of the reference's
    ``pyg_lib.ops.grouped_matmul``, pyg_lib/ops/__init__.py:99-134).  Differentiable.
    """
    needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in list(inputs) + list(others))
    if needs_grad:
        outs = list(_GroupedMatmul.apply(*(list(inputs) + list(others))))
    else:
        outs = _grouped_matmul_fwd(list(inputs), list(others))
    if biases is not None:
        for i in range(len(biases)):
            outs[i] = outs[i] + biases[i]
    return outs


# ---------------------------------------------------------------------------------------------------
# index_sort
# ---------------------------------------------------------------------------------------------------

def index_sort(
    inputs: Tensor,
    max_value: Optional[int] = None,
) -> Tuple[Tensor, Tensor]:
    """Stable ascending sort of a 1-D tensor of non-negative integers; returns ``(sorted values,
    permutation)`` and equals ``torch.sort(inputs, stable=True)`` (interface of the reference's
    ``pyg_lib.ops.index_sort``, pyg_lib/ops/__init__.py:295-321).  ``max_value`` -- any upper bound of the
    keys -- limits the number of radix passes.  The reference hands device tensors to ``torch.sort``
    (:319-320); here they run this library's LDS radix sort.
    """
    return torch.ops.pyg.index_sort(inputs, max_value)


# ---------------------------------------------------------------------------------------------------
# scatter / segment_coo / gather_coo  (pyg_lib/ops/__init__.py:353-631, 764-835)
# ---------------------------------------------------------------------------------------------------

def scatter_sum(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                dim_size: Optional[int] = None) -> Tensor:
    r"""Reduces all values from :obj:`src` into :obj:`out` at the indices specified in :obj:`index`
    along :obj:`dim`, using ``sum``.  A fresh :obj:`out` is zero-initialised; a given :obj:`out` is
    **accumulated** into (pyg_lib/ops/__init__.py:353-380)."""
    return torch.ops.pyg.scatter_sum(src, index, dim, out, dim_size)


scatter_add = scatter_sum


def scatter_mul(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                dim_size: Optional[int] = None) -> Tensor:
    r"""``mul`` reduction; a fresh :obj:`out` starts from ones, a given one is multiplied into."""
    return torch.ops.pyg.scatter_mul(src, index, dim, out, dim_size)


def scatter_mean(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                 dim_size: Optional[int] = None) -> Tensor:
    r"""``mean`` reduction (sum / count, floor division for integer dtypes; empty buckets give 0)."""
    return torch.ops.pyg.scatter_mean(src, index, dim, out, dim_size)


def scatter_min(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                dim_size: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    r"""``min`` reduction.  Returns ``(values, argindex)``; empty buckets yield value ``0`` and
    argindex ``src.size(dim)`` (sentinel); on ties the first source position wins."""
    return torch.ops.pyg.scatter_min(src, index, dim, out, dim_size)


def scatter_max(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                dim_size: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    r"""``max`` reduction.  Returns ``(values, argindex)`` (see :func:`scatter_min`)."""
    return torch.ops.pyg.scatter_max(src, index, dim, out, dim_size)


def segment_sum_coo(src: Tensor, index: Tensor, out: Optional[Tensor] = None,
                    dim_size: Optional[int] = None) -> Tensor:
    r"""Sums :obj:`src` over the runs of a **sorted** :obj:`index` along ``index.dim() - 1``."""
    return torch.ops.pyg.segment_sum_coo(src, index, out, dim_size)


segment_add_coo = segment_sum_coo


def segment_mean_coo(src: Tensor, index: Tensor, out: Optional[Tensor] = None,
                     dim_size: Optional[int] = None) -> Tensor:
    r"""Mean over the runs of a sorted :obj:`index`; buckets touched by :obj:`index` are overwritten."""
    return torch.ops.pyg.segment_mean_coo(src, index, out, dim_size)


def segment_min_coo(src: Tensor, index: Tensor, out: Optional[Tensor] = None,
                    dim_size: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    r"""Min over the runs of a sorted :obj:`index`; returns ``(values, argindex)``."""
    return torch.ops.pyg.segment_min_coo(src, index, out, dim_size)


def segment_max_coo(src: Tensor, index: Tensor, out: Optional[Tensor] = None,
                    dim_size: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    r"""Max over the runs of a sorted :obj:`index`; returns ``(values, argindex)``."""
    return torch.ops.pyg.segment_max_coo(src, index, out, dim_size)


def gather_coo(src: Tensor, index: Tensor, out: Optional[Tensor] = None) -> Tensor:
    r"""``out[..., i, ...] = src[..., index[..., i], ...]`` along ``index.dim() - 1``."""
    return torch.ops.pyg.gather_coo(src, index, out)


def scatter(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
            dim_size: Optional[int] = None, reduce: str = 'sum') -> Tensor:
    r"""Routes to the typed scatter op by :obj:`reduce` (``"sum"``/``"add"``, ``"mul"``, ``"mean"``,
    ``"min"``, ``"max"``); min/max return only the values (pyg_lib/ops/__init__.py:764-791)."""
    if reduce == 'sum' or reduce == 'add':
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == 'mul':
        return scatter_mul(src, index, dim, out, dim_size)
    if reduce == 'mean':
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == 'min':
        return scatter_min(src, index, dim, out, dim_size)[0]
    if reduce == 'max':
        return scatter_max(src, index, dim, out, dim_size)[0]
    raise ValueError(f'Unknown reduce: {reduce!r}')


def segment_coo(src: Tensor, index: Tensor, out: Optional[Tensor] = None, dim_size: Optional[int] = None,
                reduce: str = 'sum') -> Tensor:
    r"""Routes by :obj:`reduce` to the typed ``segment_*_coo`` op (pyg_lib/ops/__init__.py:794-813)."""
    if reduce == 'sum' or reduce == 'add':
        return segment_sum_coo(src, index, out, dim_size)
    if reduce == 'mean':
        return segment_mean_coo(src, index, out, dim_size)
    if reduce == 'min':
        return segment_min_coo(src, index, out, dim_size)[0]
    if reduce == 'max':
        return segment_max_coo(src, index, out, dim_size)[0]
    raise ValueError(f'Unknown reduce: {reduce!r}')


def _broadcast(index: Tensor, src: Tensor, dim: int) -> Tensor:
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def _require_float(name: str, src: Tensor) -> None:
    if not src.is_floating_point():
        raise ValueError(f'{name} requires a floating-point src tensor (got {src.dtype})')


def scatter_softmax(src: Tensor, index: Tensor, dim: int = -1, dim_size: Optional[int] = None) -> Tensor:
    r"""Softmax over the groups given by :obj:`index` (pyg_lib/ops/__init__.py:838-862): recentre by
    the per-group max, exponentiate, divide by the per-group sum."""
    _require_float('scatter_softmax', src)
    idx = _broadcast(index, src, dim)
    group_max = scatter_max(src, index, dim, dim_size=dim_size)[0]
    ex = (src - group_max.gather(dim, idx)).exp()
    group_sum = scatter_sum(ex, index, dim, dim_size=dim_size)
    return ex / group_sum.gather(dim, idx)


def scatter_log_softmax(src: Tensor, index: Tensor, dim: int = -1, dim_size: Optional[int] = None,
                        eps: float = 1e-12) -> Tensor:
    r"""Log-softmax over the groups given by :obj:`index` (pyg_lib/ops/__init__.py:865-889)."""
    _require_float('scatter_log_softmax', src)
    idx = _broadcast(index, src, dim)
    group_max = scatter_max(src, index, dim, dim_size=dim_size)[0]
    centred = src - group_max.gather(dim, idx)
    group_sum = scatter_sum(centred.exp(), index, dim, dim_size=dim_size)
    return centred - torch.log(group_sum.gather(dim, idx) + eps)


def scatter_std(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                dim_size: Optional[int] = None, unbiased: bool = True) -> Tensor:
    r"""Standard deviation per group (pyg_lib/ops/__init__.py:892-931): two :func:`scatter_sum`
    passes, Bessel's correction ``N / (N - 1)`` when :obj:`unbiased`."""
    _require_float('scatter_std', src)
    if out is not None:
        dim_size = out.size(dim)
    idx = _broadcast(index, src, dim)
    count = scatter_sum(torch.ones_like(src), idx, dim, dim_size=dim_size)
    total = scatter_sum(src, idx, dim, dim_size=dim_size)
    count_safe = count.clamp(min=1)
    dev = src - (total / count_safe).gather(dim, idx)
    res = scatter_sum(dev * dev, idx, dim, out, dim_size)
    denom = (count - 1).clamp(min=1) if unbiased else count_safe
    return (res / denom).sqrt()


def scatter_logsumexp(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                      dim_size: Optional[int] = None, eps: float = 1e-12) -> Tensor:
    r"""Numerically stable log-sum-exp per group (pyg_lib/ops/__init__.py:934-984).  Empty buckets
    give ``0`` for a fresh output and keep the caller's value when :obj:`out` is supplied."""
    _require_float('scatter_logsumexp', src)
    if out is not None:
        dim_size = out.size(dim)
    size = list(src.size())
    if dim_size is not None:
        size[dim] = dim_size
    elif index.numel() == 0:
        size[dim] = 0
    else:
        size[dim] = int(index.max().item()) + 1
    group_max = torch.full(size, float('-inf'), dtype=src.dtype, device=src.device)
    scatter_max(src, index, dim, group_max, dim_size)
    idx = _broadcast(index, src, dim)
    centred = src - group_max.gather(dim, idx)
    centred = torch.where(torch.isnan(centred), torch.full_like(centred, float('-inf')), centred)
    group_sum = scatter_sum(centred.exp(), index, dim, dim_size=dim_size)
    res = group_max + (group_sum + eps).log()
    if out is None:
        return res.nan_to_num(nan=0.0, posinf=0.0, neginf=0.0)
    keep = out.clone()
    out.copy_(torch.where(~torch.isfinite(res), keep, res))
    return out


# ---- CSR family (pyg_lib/ops/__init__.py:324-350, 634-745, 816-836) -----------------------------------
def softmax_csr(src: Tensor, ptr: Tensor, dim: int = 0) -> Tensor:
    r"""Sparsely evaluated softmax: groups the values of :obj:`src` along :obj:`dim` by the CSR
    pointer :obj:`ptr` and normalises every group on its own (pyg_lib/ops/__init__.py:324-350)."""
    dim = dim + src.dim() if dim < 0 else dim
    return torch.ops.pyg.softmax_csr(src, ptr, dim)


def segment_sum_csr(src: Tensor, indptr: Tensor, out: Optional[Tensor] = None) -> Tensor:
    r"""Row sums of :obj:`src` along ``indptr.dim() - 1`` by the CSR pointer :obj:`indptr`
    (``[..., R+1]``); a given :obj:`out` is **accumulated** into."""
    return torch.ops.pyg.segment_sum_csr(src, indptr, out)


segment_add_csr = segment_sum_csr


def segment_mean_csr(src: Tensor, indptr: Tensor, out: Optional[Tensor] = None) -> Tensor:
    r"""Row means (empty rows read 0); a given :obj:`out` is overwritten."""
    return torch.ops.pyg.segment_mean_csr(src, indptr, out)


def segment_min_csr(src: Tensor, indptr: Tensor, out: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    r"""Row minima and the source position of the first one (sentinel ``src.size(dim)`` for rows
    without a contribution)."""
    return torch.ops.pyg.segment_min_csr(src, indptr, out)


def segment_max_csr(src: Tensor, indptr: Tensor, out: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    r"""Row maxima and the source position of the first one."""
    return torch.ops.pyg.segment_max_csr(src, indptr, out)


def gather_csr(src: Tensor, indptr: Tensor, out: Optional[Tensor] = None) -> Tensor:
    r"""Inverse of :func:`segment_sum_csr`: row ``r`` of :obj:`src` is written to the positions
    ``indptr[r] .. indptr[r+1]`` of the output."""
    return torch.ops.pyg.gather_csr(src, indptr, out)


def segment_csr(src: Tensor, indptr: Tensor, out: Optional[Tensor] = None, reduce: str = 'sum') -> Tensor:
    r"""Polymorphic CSR segment dispatcher; min/max return only the value tensor."""
    if reduce == 'sum' or reduce == 'add':
        return segment_sum_csr(src, indptr, out)
    if reduce == 'mean':
        return segment_mean_csr(src, indptr, out)
    if reduce == 'min':
        return segment_min_csr(src, indptr, out)[0]
    if reduce == 'max':
        return segment_max_csr(src, indptr, out)[0]
    raise ValueError(f'Unknown reduce: {reduce!r}')


def matmul_last_variant() -> str:
    """Name of the kernel variant the last matmul call dispatched to (test/diagnostic hook)."""
    return _capi.lib().pyg_hip_matmul_last_variant().decode()


def matmul_dw_counters() -> Tuple[int, int]:
    """(specialised, general): calls served by the shape-specialised / the general-shape weight-gradient kernels since the
    library was loaded (process wide -- the backward pass runs on an autograd thread)."""
    import ctypes
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    _capi.lib().pyg_hip_matmul_dw_counters(ctypes.byref(a), ctypes.byref(b))
    return int(a.value), int(b.value)


_SCHEDULES = {'auto': 0, 'contiguous': 1, 'cyclic': 2, 'ticket': 3, 'general': 4, 'naive': 5, 'ring': 6}


def set_matmul_schedule(mode: str = 'auto') -> None:
    """Tile schedule / kernel family of the matmul calls made FROM THE CALLING THREAD (a test and measurement hook; the
    backward of a ``segment_matmul`` inherits the schedule its forward ran with).  It becomes the ``PYG_HIP_MM_SCHED_*``
    bits of the ``flags`` argument of ``pyg_hip_segment_matmul`` / ``pyg_hip_grouped_matmul`` (include/pyg_hip.h):

    * ``'auto'`` (default): ticket schedule for long relations, item ring for many short ones, contiguous ranges for
      small calls;
    * 16-bit ``K = M = 128``: ``'contiguous'`` (one tile range per workgroup), ``'cyclic'`` (every XCD sweeps its band of
      tiles), ``'ticket'`` (tiles drawn in address order from per-XCD counters, W in registers), ``'ring'`` (W slices in
      registers, X tiles through an LDS-DMA item ring) -- the same bits from each, they differ in speed only;
    * 16-bit ``K = M = 256``: ``'auto'`` / ``'ring'`` = W in registers + item ring, ``'contiguous'`` = W in LDS with 32
      rows per wave, ``'cyclic'`` / ``'ticket'`` = W in LDS with 64 rows per wave;
    * fp32 ``K = M = 128`` in split-bf16 arithmetic: ``'ring'`` forces the register-W ring kernel;
    * ``'general'`` / ``'naive'`` (measurement only): every floating-point shape through the general-shape MFMA kernel /
      every call through the one-thread-per-output kernel.

    fp32 arithmetic is not selected here: it follows ``torch.get_float32_matmul_precision()`` as in the reference
    (``'highest'``, torch's default: IEEE fp32 MFMAs; ``'high'`` / ``'medium'``: the split-bf16 kernels for
    ``K = 128, M % 128 == 0``)."""
    _capi.binding().pyg_binding_set_matmul_schedule(_SCHEDULES[mode])


@contextlib.contextmanager
def matmul_f32_split(on: bool = True):
    """Context manager over ``torch.set_float32_matmul_precision``: ``True`` -> ``'high'`` (fp32 ``K = 128, M % 128 == 0``
    matmuls run the split-bf16 kernels: three bf16 planes per operand, fp32 accumulation, HBM-bound), ``False`` ->
    ``'highest'`` (IEEE fp32 MFMAs; torch's default).  Restores the previous setting on exit.  Note that the switch is
    torch's own process-wide one -- the same the reference consults (ops/cuda/matmul_kernel.cu:158-165)."""
    prev = torch.get_float32_matmul_precision()
    torch.set_float32_matmul_precision('high' if on else 'highest')
    try:
        yield
    finally:
        torch.set_float32_matmul_precision(prev)


__all__ = [
    'grouped_matmul',
    'segment_matmul',
    'index_sort',
    'scatter',
    'scatter_sum',
    'scatter_add',
    'scatter_mul',
    'scatter_mean',
    'scatter_min',
    'scatter_max',
    'scatter_softmax',
    'scatter_log_softmax',
    'scatter_std',
    'scatter_logsumexp',
    'segment_coo',
    'segment_sum_coo',
    'segment_add_coo',
    'segment_mean_coo',
    'segment_min_coo',
    'segment_max_coo',
    'gather_coo',
    'softmax_csr',
    'segment_sum_csr',
    'segment_add_csr',
    'segment_mean_csr',
    'segment_min_csr',
    'segment_max_csr',
    'gather_csr',
    'segment_csr',
]
