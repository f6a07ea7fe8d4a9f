"""MI355X-native (gfx950) implementation of pyg-lib's hot path.

Mirrors the reference package for that path (pyg_lib/__init__.py:10-49): ``libpyg.so`` next to
this file registers the reference's ``pyg::*`` operator schemas, and ``pyg_lib_amd.ops`` /
``pyg_lib_amd.sampler`` are the same thin wrappers over ``torch.ops.pyg`` as ``pyg_lib.ops`` /
``pyg_lib.sampler``.  The operators are implemented by hand-written HIP kernels behind the C-ABI
of ``include/pyg_hip.h`` (``libpyg_hip.so``).  Device tensors never leave the device: there is no CPU fallback and no
Triton path, and without the libraries the import fails loudly.  (CPU tensors dispatch to the ``CPU`` key's plain
restatement of the same operators, csrc/binding/pyg_binding_cpu*.cpp -- host logic for tests and tooling, as the
reference registers its own CPU kernels; a device call never reaches it.)
"""
import os.path as osp

import torch

__version__ = '0.9.0+amd.r1'

_HERE = osp.dirname(osp.abspath(__file__))


def _load(name: str) -> None:
    """Registers the ``pyg::*`` operators: ``<name>.so`` is built in-tree, next to this file (the reference
    searches its package directory and only warns when the library is missing, pyg_lib/__init__.py:17-33;
    here a missing library is an error -- there is nothing to fall back to)."""
    so = osp.join(_HERE, name + '.so')
    if not osp.isfile(so):
        raise ImportError(f"pyg_lib_amd: {so} is missing.  Build it with "
                          f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950); "
                          f"there is no CPU fallback.")
    torch.ops.load_library(so)


_load('libpyg')

from pyg_lib_amd import _capi  # noqa: E402
import pyg_lib_amd.ops  # noqa: E402,F401
import pyg_lib_amd.sampler  # noqa: E402,F401


def hip_version() -> int:
    r"""Returns the HIP version :obj:`pyg_lib_amd` was compiled with (what the reference calls
    :func:`pyg_lib.cuda_version`, pyg_lib/__init__.py:43-49)."""
    return torch.ops.pyg.cuda_version()


cuda_version = hip_version

__all__ = ['__version__', 'hip_version', 'cuda_version']
