"""Importable alias of the ``nngp-src_amd/`` source directory.

The product directory is named ``nngp-src_amd`` (after the reference repository), which is not a
valid Python identifier; this shim package points its ``__path__`` at that directory so that
``import nngp_src_amd`` / ``from nngp_src_amd import stax`` resolve to ``nngp-src_amd/*.py``.
"""
import os as _os

_SRC = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nngp-src_amd")
if not _os.path.isdir(_SRC):  # pragma: no cover
    raise ImportError("nngp-src_amd/ source directory not found next to nngp_src_amd/")
__path__.insert(0, _SRC)

from . import _lib  # noqa: E402,F401  (ctypes binding of libnngp_hip.so; loads lazily)
from . import stax, predict, util  # noqa: E402,F401
from .batching import batch  # noqa: E402,F401

__all__ = ["stax", "predict", "batch", "util"]
