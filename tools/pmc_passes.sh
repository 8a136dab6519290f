#!/bin/bash
# Development: several rocprofv3 --pmc passes over a short bench run (one counter group per
# pass, kernel-trace only), CSVs under gpurun_out/pmc_<tag>_<k>/.
# usage: tools/pmc_passes.sh <tag> "<python command>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; CMD=$2
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY"
G2="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
G3="TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_WRITE_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum"
G4="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_LATENCY_sum"
k=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  k=$((k+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/gpurun_out/pmc_${TAG}_$k -o s -- $CMD > $R/gpurun_out/pmc_${TAG}_$k.log 2>&1
  tail -1 $R/gpurun_out/pmc_${TAG}_$k.log | cut -c1-200
done
