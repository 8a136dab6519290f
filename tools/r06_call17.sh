cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "policy_sampler or hipgraph" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2pi -o s -- python $GRAFT_REPO_ROOT/bench.py --workload C2pi --steps 300 --warmup 30 > $GRAFT_REPO_ROOT/gpurun_out/r06_c2pi_bench2.json 2>/dev/null
find $GRAFT_REPO_ROOT/gpurun_out/prof_c2pi -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-150
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c2pi
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r06_c2pi_bench2.json | cut -c1-600
