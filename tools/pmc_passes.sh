#!/bin/bash
# Development: several rocprofv3 --pmc passes over a short bench run (one counter group per
# pass, kernel-trace only), CSVs under gpurun_out/pmc_<tag>_<k>/.
# usage: tools/pmc_passes.sh <tag> "<python command>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; CMD=$2
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY"
G2="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
# (TCC_* / TCP_* / TA_* groups were tried once: that pass ran for ~20 GPU-minutes on this pool; SQ_* groups take seconds.)
k=0
for G in "$G1" "$G2"; do
  k=$((k+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/gpurun_out/pmc_${TAG}_$k -o s -- $CMD > $R/gpurun_out/pmc_${TAG}_$k.log 2>&1
  tail -1 $R/gpurun_out/pmc_${TAG}_$k.log | cut -c1-200
done
