#!/bin/bash
# Development: rocprofv3 SQ counter passes (kernel-trace + --pmc only) over a short bench run of one workload;
# CSVs under gpurun_out/sq_<wl>_<k>/.   usage: tools/sq_passes.sh C5 [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=$1; shift
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
G3="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_BRANCH"
# lane utilisation: thread-cycles of the VALU against its instruction-cycles; a pass of its own -- if
# a counter of it does not exist on this device only this pass is lost
G4="SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
k=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  k=$((k+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/gpurun_out/sq_${WL}_$k -o s -- \
    python $R/bench.py --workload $WL --no-cpu-baseline --steps 30 --warmup 5 "$@" > $R/gpurun_out/sq_${WL}_$k.log 2>&1
  tail -c 200 $R/gpurun_out/sq_${WL}_$k.log
done
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob("$R/gpurun_out/sq_${WL}_*/s_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "step_kernel" in k or "window_kernel" in k or "aie_jit_step" in k:
            acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob("$R/gpurun_out/sq_${WL}_*/s_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "step_kernel" in k or "window_kernel" in k or "aie_jit_step" in k:
            dur[k.split("(")[0][:60]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, d in acc.items():
    c = {name: sum(v) / len(v) for name, v in sorted(d.items())}
    insts = sum(c.get(x, 0.0) for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR",
                                        "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"))
    if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        # active lanes per vector instruction: thread-cycles / (instruction-cycles x 64 lanes)
        denom = c.get("SQ_INST_CYCLES_VALU") or c["SQ_ACTIVE_INST_VALU"]
        c["valu_active_lane_fraction"] = c["SQ_THREAD_CYCLES_VALU"] / (denom * 64.0)
    out[k] = {"counters": c, "launches_per_pass": len(next(iter(d.values()))),
              "launch_ns_under_counters_mean": (sum(dur[k]) / len(dur[k])) if dur.get(k) else None,
              "wave_instructions_per_launch": insts,
              "note": "mean per launch; rocprofv3 --kernel-trace --pmc, one counter group per pass (tools/sq_passes.sh); "
                      "cycle counters (SQ_*_CYCLES, SQ_WAIT_*, SQ_ACTIVE_*) in units of 4 clocks; launches run slower "
                      "under counter collection than in the timed bench"}
    print(k)
    for name, v in c.items():
        print("   %-26s mean %.4g" % (name, v))
try:  # which kernels were measured (bench.py flags a summary whose hash is not the running library's as stale)
    out["source_hash"] = open("$R/ai-economist_amd/csrc/libaie_hip.so.srchash").read().strip()
except OSError:
    out["source_hash"] = None
json.dump(out, open("$R/gpurun_out/r04_${WL}_sq_counters.json", "w"), indent=1)
PY
