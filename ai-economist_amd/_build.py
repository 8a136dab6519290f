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
           "aie_layout.h", "aie_glibc_math.h", "aie_glibc_tables.h"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(ROOT, "include", "aie.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-comment", "-I" + os.path.join(ROOT, "include"),
           os.path.join(CSRC, "aie_capi.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB
