"""Loads the HIP shared library behind the C ABI.  There is NO fallback: if the library
is missing or fails to load, importing the device backend raises."""
import ctypes
import os

from . import _build, _cabi

_LIBS = {}


def lib(dev=False):
    """The shipping library; dev=True: the -DAIE_DEV build with the development hooks (tools/, a few tests)."""
    if dev not in _LIBS:
        path = _build.LIB_DEV if dev else _build.LIB
        override = os.environ.get("AIE_HIP_LIBRARY")  # a prebuilt library elsewhere (deployments; A/B runs of two builds)
        if override and not dev:
            _LIBS[dev] = _cabi.bind(ctypes.CDLL(override))
            return _LIBS[dev]
        if _build.is_stale(path) and _build.have_hipcc():
            # sources newer than the binary: never run an old kernel silently (on a box without hipcc -- the GPU box
            # receives the prebuilt library -- there is nothing to rebuild with, and file times there are the copy's)
            _build.build(dev=dev)
        if not os.path.exists(path):
            raise RuntimeError(
                "HIP extension {} is not built; run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (needs hipcc). There is no CPU fallback.".format(path))
        _LIBS[dev] = _cabi.bind(ctypes.CDLL(path))
    return _LIBS[dev]
