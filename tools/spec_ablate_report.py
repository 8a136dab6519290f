#!/usr/bin/env python
"""Per-mask dynamic instruction counts of tools/spec_ablate.py's launches from a rocprofv3 --pmc run:
   python tools/spec_ablate_report.py <dir with *_counter_collection.csv> [times.json out_prefix]
With the launch times tools/spec_ablate.py wrote (ABLATE_JSON) -> <out_prefix>_c2_phase_ablation.json, _c3_...: per
switched-off phase the launch time (under counter collection) and the per-replica instruction counts."""
import json
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
times = json.load(open(sys.argv[2])) if len(sys.argv) > 3 else {}
for env in range(len(ids) // PER_ENV):
    mine = ids[env * PER_ENV:(env + 1) * PER_ENV][300:]
    print("environment", env)
    base = None
    table = []
    for k, m in enumerate(MASKS):
        chunk = mine[k * (5 + LAUNCHES) + 5:(k + 1) * (5 + LAUNCHES)]
        names = sorted(rows[chunk[0]])
        mean = {c: sum(rows[d].get(c, 0.0) for d in chunk) / len(chunk) / 4096 for c in names}
        if base is None:
            base = mean
        table.append({"mask": m, "phase_off": NAMES.get(m, "?"), "per_replica": {c.replace("SQ_", ""): round(v, 1) for c, v in mean.items()}})
        print("  %-40s " % NAMES.get(m, "?") + "  ".join("%s %7.0f (%+6.0f)" % (c.replace("SQ_INSTS_", ""), mean[c], mean[c] - base[c]) for c in names))
    if times:
        key = sorted(times, key=int)[env] if env < len(times) else None
        if key:
            for row, t in zip(table, times[key]):
                row["us_per_launch_under_counters"] = round(t["us_per_launch"], 2)
            out = "%s_c%s_phase_ablation.json" % (sys.argv[3], {"4": "2", "10": "3"}.get(key, key))
            json.dump({"workload": "gather-trade-build 25x25, %s agents, 4096 replicas, compile-time instance (traced twin), one "
                                   "phase switched off at a time from one arena snapshot (tools/spec_ablate.py under rocprofv3 "
                                   "--kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR)" % key,
                       "rows": table}, open(out, "w"), indent=1)
            print("wrote", out)
