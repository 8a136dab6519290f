#!/usr/bin/env python
"""Summarises the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into profiles/<tag>_pmc.json.

  units: both counters are in KiB (calibrated here on the 179.6 MB torch zero-fill of the
         arena, which reports WRITE_SIZE = 175 380);
  gfx950 correction: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming
         reads (MI355X_MICROARCH.md, HBM section) -> doubled.
"""
import csv
import json
import statistics as st
import sys

fetch_csv, write_csv, out = sys.argv[1:4]
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "aie_step_kernel"  # substring of the kernel name


def col(path, name, kernel=None):
    kernel = kernel or KERNEL
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and "reset" not in r["Kernel_Name"] and r["Counter_Name"] == name
            and (kernel != "aie_step_kernel" or "_log" not in r["Kernel_Name"])]


parts = KERNEL.split("+")  # "k1+k2": a step made of two launches; per-step bytes = the sum of the kernels' means
per = {k: (col(fetch_csv, "FETCH_SIZE", k), col(write_csv, "WRITE_SIZE", k)) for k in parts}
f = [sum(st.mean(per[k][0]) for k in parts)] * len(per[parts[0]][0]) if len(parts) > 1 else per[parts[0]][0]
w = [sum(st.mean(per[k][1]) for k in parts)] * len(per[parts[0]][1]) if len(parts) > 1 else per[parts[0]][1]
fill = [float(r["Counter_Value"]) for r in csv.DictReader(open(write_csv)) if "FillFunctor<unsigned char>" in r["Kernel_Name"]]
res = {
    "kernel": KERNEL,
    "per_kernel_hbm_bytes": {k: (2 * st.mean(per[k][0]) + st.mean(per[k][1])) * 1024 for k in parts},
    "launches": len(f),
    "FETCH_SIZE_KiB_mean": st.mean(f),
    "WRITE_SIZE_KiB_mean": st.mean(w),
    "read_bytes_per_launch_corrected": 2 * st.mean(f) * 1024,
    "write_bytes_per_launch": st.mean(w) * 1024,
    "hbm_bytes_per_launch": (2 * st.mean(f) + st.mean(w)) * 1024,
    "calibration_arena_fill_WRITE_SIZE_KiB": fill[0] if fill else None,
    "notes": "separate --pmc passes; KiB units; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md",
}
# which kernels were measured: the content hash of the library's sources (ai_economist_amd/_build.py) -- bench.py flags a
# summary whose hash is not the running library's as stale (round 6)
try:
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res["source_hash"] = open(os.path.join(root, "ai-economist_amd", "csrc", "libaie_hip.so.srchash")).read().strip()
except OSError:
    res["source_hash"] = None
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
