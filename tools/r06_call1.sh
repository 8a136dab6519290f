cd $GRAFT_REPO_ROOT
python tools/block_trace.py 4 > gpurun_out/r06_trace_c2.txt 2>&1
python tools/block_trace.py 4 b2b > gpurun_out/r06_trace_c2_b2b.txt 2>&1
cd /tmp && export TMPDIR=/tmp
ABLATE_JSON=$GRAFT_REPO_ROOT/gpurun_out/r06_ablate_times.json timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_ablate -o s -- python $GRAFT_REPO_ROOT/tools/spec_ablate.py 4 > $GRAFT_REPO_ROOT/gpurun_out/r06_ablate.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/spec_ablate_report.py gpurun_out/r06_ablate gpurun_out/r06_ablate_times.json gpurun_out/r06_before > gpurun_out/r06_ablate_report.txt 2>&1
rm -rf gpurun_out/r06_ablate
tail -30 gpurun_out/r06_ablate_report.txt
