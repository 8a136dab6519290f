#!/usr/bin/env python
"""Recipe for oracle/_ref/: the UNMODIFIED reference `ai_economist.foundation`, byte-compiled from the sources
where they lie under /root/reference into sourceless .pyc files (plus the map_txt layout data the scenarios read at
construction).  oracle/_ref/ is git-ignored -- no reference source enters the repository or its history -- but it
travels to the GPU box with the tree like the built .so files, so that bench.py's `cpu_baseline` leg can time the
reference's own `env.step` (base_env.py:929-1032) on the GPU node's host cores (kind "reference") and tests can use
it as a second checker.  Nothing in the product path imports it.

    python oracle/make_ref.py        (run by __graft_entry__.build() whenever /root/reference exists)
"""
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("AIE_REFERENCE_SRC", "/root/reference")
DST = os.path.join(HERE, "_ref")
DATA_DIRS = ["ai_economist/foundation/scenarios/simple_wood_and_stone/map_txt",
             "ai_economist/datasets/covid19_datasets/data_and_fitted_params"]


def main():
    pkg = os.path.join(SRC, "ai_economist")
    if not os.path.isdir(os.path.join(pkg, "foundation")):
        print("make_ref: no reference tree at %s, nothing to do" % SRC)
        return 0
    n = 0
    todo = [os.path.join(pkg, "__init__.py")]
    for root, _dirs, files in os.walk(os.path.join(pkg, "foundation")):
        todo += [os.path.join(root, f) for f in files if f.endswith(".py")]
    for src in todo:
        rel = os.path.relpath(src, SRC)
        dst = os.path.join(DST, rel + "c")  # pkg/mod.py -> pkg/mod.pyc (sourceless import)
        if os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(src):
            continue
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(src, cfile=dst, dfile=rel, doraise=True)
        n += 1
    for d in DATA_DIRS:
        os.makedirs(os.path.join(DST, d), exist_ok=True)
        for f in os.listdir(os.path.join(SRC, d)):
            if not os.path.exists(os.path.join(DST, d, f)):
                shutil.copyfile(os.path.join(SRC, d, f), os.path.join(DST, d, f))
    with open(os.path.join(DST, "PYTHON_VERSION"), "w") as f:
        f.write("%d.%d\n" % sys.version_info[:2])
    print("make_ref: %d module(s) compiled into %s" % (n, DST))
    return 0


if __name__ == "__main__":
    sys.exit(main())
