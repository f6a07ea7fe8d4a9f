"""ctypes view of the C-ABI in include/pyg_hip.h (libpyg_hip.so).

PyTorch is used here only for device memory and the current HIP stream.  There is no CPU
fallback: if the HIP library is missing, importing this module's `lib()` raises.
"""
import ctypes
import os.path as osp

import torch

_HERE = osp.dirname(osp.abspath(__file__))
_LIB = None

OK = 0
ABI_VERSION = 8  # PYG_HIP_ABI_VERSION of the include/pyg_hip.h these bindings were written against
DTYPES = {
    torch.float32: 0,
    torch.float64: 1,
    torch.float16: 2,
    torch.bfloat16: 3,
    torch.int8: 4,
    torch.uint8: 5,
    torch.int16: 6,
    torch.int32: 7,
    torch.int64: 8,
}


class Group(ctypes.Structure):
    """pyg_hip_group"""
    _fields_ = [('input', ctypes.c_void_p), ('other', ctypes.c_void_p), ('out', ctypes.c_void_p),
                ('rows', ctypes.c_int64), ('k', ctypes.c_int32), ('m', ctypes.c_int32),
                ('other_trans', ctypes.c_int32), ('reserved', ctypes.c_int32)]


def lib_path() -> str:
    return osp.join(_HERE, 'libpyg_hip.so')


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not osp.exists(path):
            raise ImportError(
                f"pyg_lib_amd: '{path}' not found. Build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(path)
        c = ctypes
        # a library built from another revision of the header would take `stream` for `flags`: refuse it
        if not hasattr(L, 'pyg_hip_abi_version') or L.pyg_hip_abi_version() != ABI_VERSION:
            got = L.pyg_hip_abi_version() if hasattr(L, 'pyg_hip_abi_version') else '< 5'
            raise ImportError(f"pyg_lib_amd: '{path}' implements C-ABI version {got}, these bindings need {ABI_VERSION}; "
                              f"rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")
        L.pyg_hip_version.restype = c.c_int64
        L.pyg_hip_last_error.restype = c.c_char_p
        L.pyg_hip_arch.restype = c.c_char_p
        L.pyg_hip_matmul_workspace_size.restype = c.c_size_t
        L.pyg_hip_matmul_workspace_size.argtypes = [c.c_int64]
        L.pyg_hip_matmul_last_variant.restype = c.c_char_p
        L.pyg_hip_segment_matmul.restype = c.c_int
        L.pyg_hip_segment_matmul.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_void_p,
                                             c.c_void_p, c.c_int64, c.c_int64, c.c_int64, c.c_int64, c.c_void_p,
                                             c.c_size_t, c.c_int, c.c_void_p]
        L.pyg_hip_grouped_matmul.restype = c.c_int
        L.pyg_hip_grouped_matmul.argtypes = [c.c_int, c.c_void_p, c.c_int64, c.c_void_p, c.c_size_t, c.c_int, c.c_void_p]
        L.pyg_hip_set_float_atomic_mode.restype = c.c_int
        L.pyg_hip_set_float_atomic_mode.argtypes = [c.c_int]
        L.pyg_hip_last_accumulate_info.restype = c.c_char_p
        L.pyg_hip_rgcn_pending_error.restype = c.c_int
        L.pyg_hip_atomic_selftest.restype = c.c_int
        L.pyg_hip_atomic_selftest.argtypes = [c.c_void_p, c.c_size_t, c.c_int, c.c_char_p, c.c_size_t, c.c_void_p]
        _LIB = L
    return _LIB


_BINDING = None


def binding() -> ctypes.CDLL:
    """libpyg.so (the torch operator library) through ctypes, for its two non-operator hooks."""
    global _BINDING
    if _BINDING is None:
        L = ctypes.CDLL(osp.join(_HERE, 'libpyg.so'))
        L.pyg_binding_set_matmul_schedule.restype = None
        L.pyg_binding_set_matmul_schedule.argtypes = [ctypes.c_int]
        L.pyg_binding_get_matmul_schedule.restype = ctypes.c_int
        _BINDING = L
    return _BINDING


def check(rc: int) -> None:
    if rc != OK:
        raise RuntimeError(lib().pyg_hip_last_error().decode())


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"pyg_lib_amd: '{name}' must live on a HIP device (got {t.device}); this build has no CPU path")
