"""MI355X-native (gfx950) implementation of pyg-lib's hot path.

Mirrors the reference package layout for that path (pyg_lib/__init__.py:10,38-49):
``pyg_lib_amd.ops`` and ``pyg_lib_amd.sampler`` expose the same functions, arguments and defaults
as ``pyg_lib.ops`` / ``pyg_lib.sampler``; the kernels live in ``libpyg_hip.so`` (C-ABI,
include/pyg_hip.h).  There is no CPU path and no Triton path.
"""
from pyg_lib_amd import _capi

__version__ = '0.9.0+amd.r1'


def hip_version() -> int:
    r"""Returns the HIP version :obj:`pyg_lib_amd` was compiled with (the counterpart of
    :func:`pyg_lib.cuda_version`, pyg_lib/__init__.py:43-49)."""
    return int(_capi.lib().pyg_hip_version())


# drop-in spelling used by PyG's version checks
cuda_version = hip_version

import pyg_lib_amd.ops  # noqa: E402,F401
import pyg_lib_amd.sampler  # noqa: E402,F401

__all__ = ['__version__', 'hip_version', 'cuda_version']
