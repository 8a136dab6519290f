#!/usr/bin/env python
import collections
import csv
import sys

MASKS = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 63]
NAMES = {0: "full", 1: "-components", 2: "-regen", 4: "-spatial", 8: "-flat+masks", 16: "-rewards", 32: "-store",
         64: "-flat stageA", 128: "-flat cda", 256: "-flat tax", 512: "-masks", 1024: "-copy-out", 2048: "-build",
         4096: "-cda", 8192: "-gather", 16384: "-tax", 63: "base only"}
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("aie_step_kernel"):
        rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for name, v in rows.items():
    v.sort()
    vals = [x for _, x in v][-20 * len(MASKS):]
    per = [sum(vals[i * 20 + 5:(i + 1) * 20]) / 15 / 4096 for i in range(len(MASKS))]
    print(name)
    for m, x in zip(MASKS, per):
        print("   %-14s %8.0f  (delta %+7.0f)" % (NAMES[m], x, x - per[0]))
