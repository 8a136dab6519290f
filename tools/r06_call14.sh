cd $GRAFT_REPO_ROOT
SQ=1 bash tools/profile_round6.sh C4x C4xu C5 > gpurun_out/r06_profile_b.txt 2>&1
tail -12 gpurun_out/r06_profile_b.txt
cd /tmp && export TMPDIR=/tmp
ABLATE_JSON=$GRAFT_REPO_ROOT/gpurun_out/r06_ablate_times_after.json timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_ablate -o s -- python $GRAFT_REPO_ROOT/tools/spec_ablate.py 4 > $GRAFT_REPO_ROOT/gpurun_out/r06_ablate_after.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/spec_ablate_report.py gpurun_out/r06_ablate gpurun_out/r06_ablate_times_after.json gpurun_out/r06_after > gpurun_out/r06_ablate_report_after.txt 2>&1
rm -rf gpurun_out/r06_ablate
python tools/spec_ablate.py 4 > gpurun_out/r06_ablate_times_clean_after.txt 2>&1
tail -32 gpurun_out/r06_ablate_report_after.txt | cut -c1-200
