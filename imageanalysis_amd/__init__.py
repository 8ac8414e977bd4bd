"""MI355X-native implementation of imageAnalysisLib's per-pixel path.

The product is the C-ABI library ``imageanalysis_amd/lib/libiak.so`` (declared in
``include/iak.h``) and the C++ API layer above it.  This Python package is a thin
ctypes loader plus torch-tensor conveniences used by the tests and ``bench.py``; it does
no image arithmetic of its own and raises if the HIP library is missing.
"""
from . import lib  # noqa: F401

__all__ = ["lib"]
