cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/c1/gputests.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c1/bench_default.json 2> gpurun_out/c1/bench_default.err
timeout 300 python tools/spec_ablate.py 4 10 > gpurun_out/c1/ablate.txt 2>&1
timeout 120 python tools/block_trace.py 4 b2b > gpurun_out/c1/trace4.txt 2>&1
timeout 120 python tools/block_trace.py 10 b2b > gpurun_out/c1/trace10.txt 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/c1/abl_pmc -o s -- python $R/tools/spec_ablate.py 4 10 > $R/gpurun_out/c1/ablate_pmc.txt 2>&1
cd $R
python tools/spec_ablate_report.py gpurun_out/c1/abl_pmc > gpurun_out/c1/ablate_insts.txt 2>&1
rm -rf gpurun_out/c1/abl_pmc
tail -3 gpurun_out/c1/gputests.txt; cat gpurun_out/c1/ablate.txt | head -50
