#!/usr/bin/env python
"""Per-mask dynamic instruction counts of tools/spec_ablate.py's launches from a rocprofv3 --pmc run:
   python tools/spec_ablate_report.py <dir with *_counter_collection.csv> [envs]"""
import collections
import csv
import glob
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from spec_ablate import LAUNCHES, MASKS, NAMES  # noqa: E402

PER_ENV = 300 + len(MASKS) * (5 + LAUNCHES)
rows = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
print("step-kernel dispatches:", len(ids), "expected per environment:", PER_ENV)
for env in range(len(ids) // PER_ENV):
    mine = ids[env * PER_ENV:(env + 1) * PER_ENV][300:]
    print("environment", env)
    base = None
    for k, m in enumerate(MASKS):
        chunk = mine[k * (5 + LAUNCHES) + 5:(k + 1) * (5 + LAUNCHES)]
        names = sorted(rows[chunk[0]])
        mean = {c: sum(rows[d].get(c, 0.0) for d in chunk) / len(chunk) / 4096 for c in names}
        if base is None:
            base = mean
        print("  %-40s " % NAMES.get(m, "?") + "  ".join("%s %7.0f (%+6.0f)" % (c.replace("SQ_INSTS_", ""), mean[c], mean[c] - base[c]) for c in names))
