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
LIB_DEV = os.path.join(CSRC, "libaie_hip_dev.so")
SOURCES = ["aie_capi.hip", "aie_kernels.hip", "aie_kernels_ose.hip", "aie_kernels_saez.hip", "aie_kernels_covid.hip",
           "aie_layout.h", "aie_glibc_math.h", "aie_glibc_tables.h", "aie_spec_generated.h", "aie_jit.h"]


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
    h.update(b"recipe 3: -fvisibility=hidden + version script\n")
    for d in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(ROOT, "include", "aie.h")]:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale(lib=None):
    """True when the library is missing or was built from other sources: judged by a content hash kept beside the
    binary (file times do not survive the copy to the GPU box)."""
    lib = lib or LIB
    if not os.path.exists(lib) or not os.path.exists(lib + ".srchash"):
        return True
    with open(lib + ".srchash") as f:
        return f.read().strip() != _source_hash()


class _BuildLock:
    """One builder at a time per tree: every rank of a multi-process launch may find the library stale at the same
    moment.  flock on a file beside the sources; the others wait, then find the library fresh."""

    def __init__(self, name=""):
        self.name = name

    def __enter__(self):
        import fcntl

        self.f = open(os.path.join(CSRC, ".build%s.lock" % self.name), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl

        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def declared_symbols():
    """The functions include/aie.h declares = the library's whole export list."""
    import re

    with open(os.path.join(ROOT, "include", "aie.h")) as f:
        return sorted(set(re.findall(r"\b(aie_[a-z_]+)\s*\(", f.read())) - {"aie_env"})


def _compile(lib, extra, verbose):
    # compile to a temporary name and rename: a reader never dlopens a half-written file
    tmp = "%s.tmp%d" % (lib, os.getpid())
    # export list = the header (kernel handles are forced to default visibility by the HIP front end, so
    # -fvisibility=hidden alone does not keep them out of the dynamic symbol table; a linker version script does)
    vmap = lib + ".map"
    with open(vmap, "w") as f:
        f.write("{\n  global:\n")
        for sym in declared_symbols() + (["aie_dev_*", "aie_test_glibc_math"] if "-DAIE_DEV" in extra else []):
            f.write("    %s;\n" % sym)
        f.write("  local: *;\n};\n")
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
           "-Wl,--version-script=" + vmap, "-Wno-comment",
           "-I" + os.path.join(ROOT, "include")] + extra + [os.path.join(CSRC, "aie_capi.hip"), "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, lib)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    with open(lib + ".srchash.tmp%d" % os.getpid(), "w") as f:
        f.write(_source_hash() + "\n")
    os.replace(lib + ".srchash.tmp%d" % os.getpid(), lib + ".srchash")


def build(force=False, verbose=False, dev=False):
    """Builds (if stale) and returns the shipping library; dev=True: the -DAIE_DEV variant with the development hooks
    (aie_dev_*, aie_test_glibc_math, traced kernels) that tools/ and a few tests load -- never the product path."""
    # the constant parameter images of the compile-time step-kernel instances follow the layout code
    from . import _specs

    lib = LIB_DEV if dev else LIB
    with _BuildLock():
        if _specs.write_header() and verbose:
            print("regenerated", _specs.OUT)
    with _BuildLock("_dev" if dev else "_lib"):
        if not force and not is_stale(lib):
            return lib
        _compile(lib, ["-DAIE_DEV"] if dev else [], verbose)
    return lib
