"""Host-side mirror of ``pyg_lib.ops`` for the hot path (pyg_lib/ops/__init__.py).

Same names, argument meaning, defaults and error behaviour as the reference; every op runs a
hand-written gfx950 kernel through the C-ABI of include/pyg_hip.h.  Tensors must live on a HIP
device -- there is no CPU kernel and no Triton path in this package.
"""
from typing import List, Optional, Tuple

import ctypes

import torch
from torch import Tensor

from pyg_lib_amd import _capi


def _workspace(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _dtype_code(t: Tensor) -> int:
    if t.dtype not in _capi.DTYPES:
        raise RuntimeError(f'pyg_lib_amd: unsupported dtype {t.dtype}')
    return _capi.DTYPES[t.dtype]


# ---------------------------------------------------------------------------------------------------
# segment_matmul
# ---------------------------------------------------------------------------------------------------

def _segment_matmul_fwd(inputs: Tensor, ptr: Tensor, other: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    # argument checks of the operator front, pyg_lib/csrc/ops/matmul.cpp:41-61
    if inputs.dtype != other.dtype:
        raise RuntimeError(f"segment_matmul: expected 'input' and 'other' to have the same dtype, but got "
                           f"{inputs.dtype} and {other.dtype}")
    if inputs.dim() != 2:
        raise RuntimeError(f"segment_matmul: expected 2-dimensional 'input', got {inputs.dim()} dims")
    if ptr.dim() != 1:
        raise RuntimeError(f"segment_matmul: expected 1-dimensional 'ptr', got {ptr.dim()} dims")
    if other.dim() != 3:
        raise RuntimeError(f"segment_matmul: expected 3-dimensional 'other', got {other.dim()} dims")
    if other.size(1) != inputs.size(-1):
        raise RuntimeError(f"segment_matmul: expected 'other' to have size {inputs.size(-1)} at dimension 1, "
                           f"but got {other.size(1)}")
    if ptr.numel() != other.size(0) + 1:
        raise RuntimeError(f"segment_matmul: expected 'ptr' to have {other.size(0) + 1} elements, "
                           f"but got {ptr.numel()}")
    if ptr.dtype != torch.int64:
        # the reference reads size.data_ptr<int64_t>() (matmul_kernel.cpp:414)
        raise RuntimeError('segment_matmul: expected scalar type Long for ptr')
    _capi.require_device(inputs, 'inputs')
    _capi.require_device(other, 'other')
    if other.device != inputs.device:
        raise RuntimeError("segment_matmul: 'inputs' and 'other' must be on the same device")

    L = _capi.lib()
    x = inputs.contiguous()
    w = other.contiguous()
    N, K = x.shape
    B, _, M = w.shape
    out = x.new_empty((N, M))
    if ptr.is_cuda:
        p = ptr.contiguous()
        on_dev = 1
    else:
        p = ptr.contiguous()
        on_dev = 0
    b = None
    if bias is not None:
        b = bias.to(dtype=x.dtype, device=x.device).contiguous()
        if b.shape != (B, M):
            raise RuntimeError(f"segment_matmul: expected 'bias' of shape [{B}, {M}], got {list(b.shape)}")
    with torch.cuda.device(x.device):
        ws = _workspace(L.pyg_hip_matmul_workspace_size(B), x.device)
        rc = L.pyg_hip_segment_matmul(_dtype_code(x), x.data_ptr(), p.data_ptr(), on_dev, w.data_ptr(),
                                      b.data_ptr() if b is not None else None, out.data_ptr(), N, K, M, B,
                                      ws.data_ptr(), ws.numel(), _capi.stream_ptr(x.device))
    _capi.check(rc)
    # keep the staging tensors alive until the stream has consumed them
    for t in (ws, p, x, w) + ((b,) if b is not None else ()):
        if t.is_cuda:
            t.record_stream(torch.cuda.current_stream(x.device))
    return out


class _SegmentMatmul(torch.autograd.Function):
    """Mirrors SegmentMatmul (pyg_lib/csrc/ops/autograd/matmul_kernel.cpp:68-111)."""
    @staticmethod
    def forward(ctx, inputs: Tensor, ptr: Tensor, other: Tensor) -> Tensor:
        ctx.save_for_backward(inputs, ptr, other)
        return _segment_matmul_fwd(inputs, ptr, other)

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        inputs, ptr, other = ctx.saved_tensors
        grad_in = grad_other = None
        if ctx.needs_input_grad[0]:
            # dX = segment_matmul(dY, ptr, W^T)   (:86-90)
            grad_in = _segment_matmul_fwd(grad_out, ptr, other.transpose(-2, -1))
        if ctx.needs_input_grad[2]:
            # dW[b] = X_b^T @ dY_b, stacked      (:92-107)
            sizes = (ptr[1:] - ptr[:-1]).cpu().tolist()
            xs = inputs.split(sizes, dim=0)
            gs = grad_out.split(sizes, dim=0)
            grad_other = torch.stack([x.t() @ g for x, g in zip(xs, gs)], dim=0)
        return grad_in, None, grad_other


def segment_matmul(
    inputs: Tensor,
    ptr: Tensor,
    other: Tensor,
    bias: Optional[Tensor] = None,
) -> Tensor:
    r"""Performs dense-dense matrix multiplication according to segments along
    the first dimension of :obj:`inputs` as given by :obj:`ptr`
    (same contract as :func:`pyg_lib.ops.segment_matmul`, pyg_lib/ops/__init__.py:137-172).

    .. code-block:: python

        inputs = torch.randn(8, 16, device='cuda')
        ptr = torch.tensor([0, 5, 8])
        other = torch.randn(2, 16, 32, device='cuda')

        out = pyg_lib_amd.ops.segment_matmul(inputs, ptr, other)
        assert out.size() == (8, 32)
        assert out[0:5] == inputs[0:5] @ other[0]
        assert out[5:8] == inputs[5:8] @ other[1]

    Args:
        inputs: The left operand 2D matrix of shape :obj:`[N, K]`.
        ptr: Compressed vector of shape :obj:`[B + 1]`, holding the boundaries
            of segments. May live on the host or on the device; neither
            placement synchronises.
        other: The right operand 3D tensor of shape :obj:`[B, K, M]`.
        bias: The bias term of shape :obj:`[B, M]`.

    Returns:
        The 2D output matrix of shape :obj:`[N, M]`.
    """
    needs_grad = torch.is_grad_enabled() and (inputs.requires_grad or other.requires_grad)
    if not needs_grad:
        # bias is a fused epilogue of the kernel (reference: B python-side slice adds, :169-171)
        return _segment_matmul_fwd(inputs, ptr, other, bias)
    out = _SegmentMatmul.apply(inputs, ptr, other)
    if bias is not None:
        for i in range(ptr.numel() - 1):
            out[ptr[i]:ptr[i + 1]] += bias[i]
    return out


# ---------------------------------------------------------------------------------------------------
# grouped_matmul
# ---------------------------------------------------------------------------------------------------

def _grouped_matmul_fwd(inputs: List[Tensor], others: List[Tensor]) -> List[Tensor]:
    # argument checks of the operator front, pyg_lib/csrc/ops/matmul.cpp:12-38
    if len(inputs) != len(others):
        raise RuntimeError("Number of 'input' tensors must match number of 'other' tensors")
    if len(inputs) == 0:
        return []
    dt = inputs[0].dtype
    for i, (a, o) in enumerate(zip(inputs, others)):
        if a.dtype != dt or o.dtype != dt:
            raise RuntimeError(f'grouped_matmul: expected all tensors to have dtype {dt} (group {i})')
        if a.dim() != 2 or o.dim() != 2:
            raise RuntimeError(f'grouped_matmul: expected 2-dimensional tensors (group {i})')
        if o.size(0) != a.size(-1):
            raise RuntimeError(f"grouped_matmul: expected 'other[{i}]' to have size {a.size(-1)} at dimension 0, "
                               f"but got {o.size(0)}")
        _capi.require_device(a, f'inputs[{i}]')
        _capi.require_device(o, f'others[{i}]')
    dev = inputs[0].device
    L = _capi.lib()
    G = len(inputs)
    groups = (_capi.Group * G)()
    keep = []
    outs = []
    for i, (a, o) in enumerate(zip(inputs, others)):
        a_c = a.contiguous()
        # a transposed view ([K, M] with strides (1, K)) is read in place by the kernel
        trans = 0
        if o.is_contiguous():
            o_c = o
        elif o.t().is_contiguous():
            o_c, trans = o, 1
        else:
            o_c = o.contiguous()
        out = a_c.new_empty((a_c.size(0), o.size(-1)))
        groups[i] = _capi.Group(a_c.data_ptr(), o_c.data_ptr(), out.data_ptr(), a_c.size(0), a_c.size(1),
                                o.size(-1), trans, 0)
        keep += [a_c, o_c]
        outs.append(out)
    with torch.cuda.device(dev):
        ws = _workspace(L.pyg_hip_matmul_workspace_size(G), dev)
        rc = L.pyg_hip_grouped_matmul(_capi.DTYPES[dt], ctypes.byref(groups), G, ws.data_ptr(), ws.numel(),
                                      _capi.stream_ptr(dev))
    _capi.check(rc)
    for t in keep + [ws]:
        t.record_stream(torch.cuda.current_stream(dev))
    return outs


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
    r"""Performs dense-dense matrix multiplication according to groups
    (same contract as :func:`pyg_lib.ops.grouped_matmul`, pyg_lib/ops/__init__.py:99-134).

    Args:
        inputs: List of left operand 2D matrices of shapes :obj:`[N_i, K_i]`.
        others: List of right operand 2D matrices of shapes :obj:`[K_i, M_i]`.
        biases: Optional bias terms to apply for each element.

    Returns:
        List of 2D output matrices of shapes :obj:`[N_i, M_i]`.
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


def matmul_last_variant() -> str:
    """Name of the kernel variant the last matmul call dispatched to (test/diagnostic hook)."""
    return _capi.lib().pyg_hip_matmul_last_variant().decode()


__all__ = [
    'grouped_matmul',
    'segment_matmul',
]
