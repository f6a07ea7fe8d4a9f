"""Diagnostics of the accumulate paths that still use atomics (include/pyg_hip.h, "Floating-point accumulation").

The weight gradients, large scatter / COO sums and the CSR family are atomic-free and bit-reproducible.  Small or
element-wise indexed ``scatter_sum``, float64 sums and the fused R-GCN layer add through hardware floating-point atomics;
:func:`set_float_atomic_mode` switches them to compare-and-swap loops, :func:`atomic_selftest` checks the underlying
clear -> add -> read-back pattern on this process's own memory and stream, :func:`last_accumulate_info` says what the
last such launch was.  tests/conftest.py uses all three to diagnose a failing test on the spot.
"""
import ctypes
from typing import Tuple

import torch

from pyg_lib_amd import _capi

_MODES = {'hw': 0, 'cas': 1}


def set_float_atomic_mode(mode: str) -> str:
    r"""``'hw'`` (hardware floating-point atomic adds, the default) or ``'cas'`` (compare-and-swap loops) for every kernel
    that still accumulates through atomics; process-wide default, also settable through ``PYG_HIP_FLOAT_ATOMICS``.
    Returns the previous mode."""
    before = _capi.lib().pyg_hip_set_float_atomic_mode(_MODES[mode])
    return 'cas' if before else 'hw'


def last_accumulate_info() -> str:
    return _capi.lib().pyg_hip_last_accumulate_info().decode()


def atomic_selftest(rounds: int = 4, device=None, megabytes: int = 16) -> Tuple[int, str]:
    r"""Runs ``pyg_hip_atomic_selftest`` on a block of the caching allocator and the current stream.  Returns (number of
    failing variants out of 30, report text)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(device):
        scratch = torch.empty(megabytes << 20, dtype=torch.uint8, device=device)
        report = ctypes.create_string_buffer(16384)
        rc = _capi.lib().pyg_hip_atomic_selftest(scratch.data_ptr(), scratch.numel(), rounds, report, len(report),
                                                 _capi.stream_ptr(device))
        if rc < 0:
            _capi.check(rc)
    return rc, report.value.decode()
