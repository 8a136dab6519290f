"""Builds the HIP shared library in-tree (ai-economist_amd/csrc/libaie_hip.so).

hipcc cross-compiles for gfx950 without a GPU; the built .so is git-ignored but travels
with the source tree to the GPU box.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libaie_hip.so")
SOURCES = ["aie_capi.hip", "aie_kernels.hip", "aie_kernels_ose.hip", "aie_kernels_saez.hip", "aie_kernels_covid.hip",
           "aie_layout.h", "aie_glibc_math.h", "aie_glibc_tables.h", "aie_spec_generated.h"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")


def have_hipcc():
    try:
        hipcc_path()
        return True
    except RuntimeError:
        return False


def _source_hash():
    import hashlib

    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(ROOT, "include", "aie.h")]:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale():
    """True when the library is missing or was built from other sources: judged by a content hash kept beside the
    binary (file times do not survive the copy to the GPU box)."""
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".srchash"):
        return True
    with open(LIB + ".srchash") as f:
        return f.read().strip() != _source_hash()


def build(force=False, verbose=False):
    # the constant parameter images of the compile-time step-kernel instances follow the layout code
    from . import _specs

    if _specs.write_header() and verbose:
        print("regenerated", _specs.OUT)
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-comment", "-I" + os.path.join(ROOT, "include"),
           os.path.join(CSRC, "aie_capi.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(LIB + ".srchash", "w") as f:
        f.write(_source_hash() + "\n")
    return LIB
