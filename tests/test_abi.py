"""The C-ABI library loads and exports every symbol include/pyg_hip.h declares (no GPU needed)."""
import ctypes
import os.path as osp
import re

import pytest

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


def declared_symbols():
    text = open(osp.join(ROOT, 'include', 'pyg_hip.h')).read()
    return sorted(set(re.findall(r'PYG_HIP_API\s+[\w\s\*]+?\b(pyg_hip_\w+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert 'pyg_hip_segment_matmul' in syms and 'pyg_hip_grouped_matmul' in syms
    assert len(syms) >= 8


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(osp.join(ROOT, 'pyg_lib_amd', 'libpyg_hip.so'))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f'not exported: {missing}'
    lib.pyg_hip_arch.restype = ctypes.c_char_p
    assert lib.pyg_hip_arch() == b'gfx950'
    lib.pyg_hip_version.restype = ctypes.c_int64
    assert lib.pyg_hip_version() > 0


def test_dtype_codes_match_between_header_oracle_and_python():
    import torch
    import oracle
    from pyg_lib_amd import _capi
    text = open(osp.join(ROOT, 'include', 'pyg_hip.h')).read()
    codes = dict((k, int(v)) for k, v in re.findall(r'(PYG_[FIU]\d+|PYG_BF16)\s*=\s*(\d+)', text))
    assert codes['PYG_F32'] == oracle.F32 == _capi.DTYPES[torch.float32]
    assert codes['PYG_F64'] == oracle.F64 == _capi.DTYPES[torch.float64]
    assert codes['PYG_F16'] == oracle.F16 == _capi.DTYPES[torch.float16]
    assert codes['PYG_BF16'] == oracle.BF16 == _capi.DTYPES[torch.bfloat16]
    assert codes['PYG_I64'] == _capi.DTYPES[torch.int64]


def test_every_operator_family_has_a_cpu_key():
    import torch
    import pyg_lib_amd  # noqa: F401
    # CPU is listed for matmul / index_sort / scatter / coo (+ the CSR family and the samplers); parity of the
    # CPU kernels is tests/test_cpu_key.py.  A device tensor never reaches them (the dispatcher picks by device).
    for op in ('segment_matmul', 'grouped_matmul', 'index_sort', 'neighbor_sample', 'scatter_sum', 'scatter_mul',
               'scatter_min', 'scatter_max', 'segment_sum_coo', 'segment_mean_coo', 'segment_min_coo', 'segment_max_coo',
               'gather_coo', 'segment_sum_csr', 'segment_mean_csr', 'segment_min_csr', 'segment_max_csr', 'gather_csr',
               'softmax_csr', 'softmax_csr_backward'):
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f'pyg::{op}', 'CPU'), op


def test_product_never_imports_the_oracle():
    import subprocess
    out = subprocess.run(['grep', '-rIl', '-E', r'(^|\s)(import oracle|from oracle)|liboracle', 
                          osp.join(ROOT, 'pyg_lib_amd')], capture_output=True, text=True).stdout.strip()
    assert out == '', f'product references the oracle: {out}'
