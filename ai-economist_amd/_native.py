"""Loads the HIP shared library behind the C ABI.  There is NO fallback: if the library
is missing or fails to load, importing the device backend raises."""
import ctypes
import os

from . import _build, _cabi

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_build.LIB):
            raise RuntimeError(
                "HIP extension {} is not built; run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (needs hipcc). There is no CPU fallback.".format(_build.LIB))
        if _build.is_stale() and _build.have_hipcc():
            # sources newer than the binary: never run an old kernel silently (on a box without hipcc -- the GPU box
            # receives the prebuilt library -- there is nothing to rebuild with, and file times there are the copy's)
            _build.build()
        _LIB = _cabi.bind(ctypes.CDLL(_build.LIB))
    return _LIB
