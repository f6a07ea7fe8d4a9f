"""MI355X-native (gfx950) implementation of pyg-lib's hot path.

Mirrors the reference package for that path (pyg_lib/__init__.py:10-49): ``libpyg.so`` next to
this file registers the reference's ``pyg::*`` operator schemas, and ``pyg_lib_amd.ops`` /
``pyg_lib_amd.sampler`` are the same thin wrappers over ``torch.ops.pyg`` as ``pyg_lib.ops`` /
``pyg_lib.sampler``.  The operators are implemented by hand-written HIP kernels behind the C-ABI
of ``include/pyg_hip.h`` (``libpyg_hip.so``).  There is no CPU path and no Triton path: without
the libraries the import fails loudly.
"""
import importlib.machinery
import os.path as osp

import torch

__version__ = '0.9.0+amd.r1'


def load_library(lib_name: str) -> None:
    # same discovery as pyg_lib/__init__.py:17-33, but a missing library is an error, not a warning
    loader_details = (
        importlib.machinery.ExtensionFileLoader,
        importlib.machinery.EXTENSION_SUFFIXES,
    )
    path = osp.dirname(osp.abspath(__file__))
    ext_finder = importlib.machinery.FileFinder(path, loader_details)
    spec = ext_finder.find_spec(lib_name)
    if spec is None:
        raise ImportError(
            f"pyg_lib_amd: could not find shared library '{lib_name}' in {path}. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950); "
            f"there is no CPU fallback.")
    torch.ops.load_library(spec.origin)


load_library('libpyg')

from pyg_lib_amd import _capi  # noqa: E402
import pyg_lib_amd.ops  # noqa: E402,F401
import pyg_lib_amd.sampler  # noqa: E402,F401


def hip_version() -> int:
    r"""Returns the HIP version :obj:`pyg_lib_amd` was compiled with (what the reference calls
    :func:`pyg_lib.cuda_version`, pyg_lib/__init__.py:43-49)."""
    return torch.ops.pyg.cuda_version()


cuda_version = hip_version

__all__ = ['__version__', 'hip_version', 'cuda_version']
