"""debigulator_amd -- MI355X-native batched DEFLATE inflate / PNG de-filter path behind
debigulator's inflate()/decode_png()/decode_gz() header API.

Native code lives in csrc/ (HIP kernels + C-ABI shim, see include/*.h); this package is
the thin host-side mirror used by tests and bench.py.
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
