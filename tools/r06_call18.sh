cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "policy_sampler or hipgraph" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for W in 0 1 2; do
AIE_SAMPLER_WAVES_LOG2=$W timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2pi -o s -- python $GRAFT_REPO_ROOT/bench.py --workload C2pi --steps 300 --warmup 30 > $GRAFT_REPO_ROOT/gpurun_out/r06_c2pi_bench_w$W.json 2>/dev/null
echo "waves log2 $W"; find $GRAFT_REPO_ROOT/gpurun_out/prof_c2pi -name "*kernel_stats.csv" | head -1 | xargs -r grep -E "sample_policy|step_kernel" | cut -c1-150
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c2pi
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r06_c2pi_bench_w$W.json | cut -c1-300
done
