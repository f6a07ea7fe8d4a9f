"""Host-side mirror of ``pyg_lib.ops`` for the hot path (pyg_lib/ops/__init__.py).

Same names, argument meaning, defaults and error behaviour as the reference; every op runs a
hand-written gfx950 kernel through the C-ABI of include/pyg_hip.h.  Tensors must live on a HIP
device -- there is no CPU kernel and no Triton path in this package.
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from pyg_lib_amd import _capi


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
