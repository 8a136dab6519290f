#!/bin/bash
# Development (round 5): SQ counters of aie_sample_policy_actions_kernel over a short C2pi run (two --pmc passes,
# kernel-trace only) -> means per launch.   usage (GPU box): tools/sq_sampler.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD"; do
  k=$((k+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/gpurun_out/sqs_$k -o s -- python $R/bench.py --workload C2pi --steps 40 --warmup 10 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$R/gpurun_out/sqs_*/s_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "sample_policy" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-26s mean %.4g (n %d)" % (k, sum(v) / len(v), len(v)))
PY
rm -rf $R/gpurun_out/sqs_*
