#!/usr/bin/env python
"""Development (round 5, VERDICT r4 #7 -- one bounded experiment): the build-time instance and the run-time (hiprtc)
instance of the SAME family (BASELINE configs[1]) in ONE process, 200 alternating launches.

   python tools/jit_gap_experiment.py run                 the launches (run it under `rocprofv3 --kernel-trace --hip-trace
                                                          --output-format csv -d <dir> -o s --`)
   python tools/jit_gap_experiment.py report <dir>        per kernel: duration, end-of-previous-kernel -> start, launch API
                                                          call -> start, from the two traces
GPU only.  AIE_JIT_CACHE points the run-time instance's code object at a directory that travels back (descriptor diff:
llvm-readelf --notes on it and on the library's bundle)."""
import csv
import glob
import os
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def run():
    import torch

    import bench
    from helpers import make_env

    envs = []
    for forced in (False, True):
        env = make_env(dict(bench.C2_CFG), n_envs=4096, device="cuda:0")
        env.seed(1)
        env.reset()
        be = env.backend
        if forced:
            os.environ["AIE_JIT_FORCE"] = "1"
            be.lib.aie_select_step_kernel(be.handle, 1)
            assert env.specialize(), be.lib.aie_last_error(be.handle)
            be.lib.aie_select_step_kernel(be.handle, 0)
            assert be.lib.aie_step_kernel_instance(be.handle) == 1000
        else:
            assert be.lib.aie_step_kernel_instance(be.handle) == 0
        envs.append(be)
    acts = [be.sample_random_actions(5, 0, slot=0) for be in envs]
    for _ in range(100):
        for be, (a, p) in zip(envs, acts):
            be.step(a, p)
    torch.cuda.synchronize()
    for _ in range(200):  # alternating, back to back
        for be, (a, p) in zip(envs, acts):
            be.step(a, p)
    torch.cuda.synchronize()
    # and in blocks of 50 of the same kernel (what a rollout looks like)
    for rep in range(2):
        for be, (a, p) in zip(envs, acts):
            for _ in range(50):
                be.step(a, p)
    torch.cuda.synchronize()


def report(d):
    kt = [f for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
    ht = [f for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)][0]
    ks = sorted(({"name": r["Kernel_Name"].split("(")[0][:40], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]),
                  "cid": r.get("Correlation_Id")} for r in csv.DictReader(open(kt))), key=lambda r: r["s"])
    api = {r.get("Correlation_Id"): (r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(ht))
           if "Launch" in r["Function"]}
    steps = [k for k in ks if "step" in k["name"]]
    n = len(steps)
    alt, blocks = steps[n - 500 - 0: n - 200], steps[n - 200:]   # the 400 alternating launches (last 300 of them), the 4 x 50 blocks
    for label, seq in (("alternating A B A B", alt), ("blocks of 50", blocks)):
        print(label)
        by = {}
        for prev, k in zip(seq, seq[1:]):
            rec = by.setdefault(k["name"], {"dur": [], "gap": [], "api": [], "after_same": []})
            rec["dur"].append((k["e"] - k["s"]) / 1e3)
            rec["gap"].append((k["s"] - prev["e"]) / 1e3)
            if prev["name"] == k["name"]:
                rec["after_same"].append((k["s"] - prev["e"]) / 1e3)
            a = api.get(k["cid"])
            if a:
                rec["api"].append((k["s"] - a[2]) / 1e3)
        for name, r in by.items():
            print("  %-42s n %3d  duration %.2f us (min %.2f)  prev end -> start %.2f us  launch call returned -> start %.1f us%s" % (
                name, len(r["dur"]), st.mean(r["dur"]), min(r["dur"]), st.median(r["gap"]),
                st.median(r["api"]) if r["api"] else float("nan"),
                "  (after its own kind: %.2f)" % st.median(r["after_same"]) if r["after_same"] else ""))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
