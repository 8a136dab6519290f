cd $GRAFT_REPO_ROOT
SQ=1 bash tools/profile_round6.sh C3 > gpurun_out/r06_profile_c3.txt 2>&1
tail -1 gpurun_out/summ/r06_c3_bench.json | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed" > gpurun_out/r06_final_gpu_tests.txt
cat gpurun_out/r06_final_gpu_tests.txt
