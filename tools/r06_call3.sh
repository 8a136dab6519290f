cd $GRAFT_REPO_ROOT
bash tools/ab2.sh "--workload C2 --steps 2000 --warmup 100" 2 base v2 > gpurun_out/r06_ab_v2.txt 2>&1
cat gpurun_out/r06_ab_v2.txt
timeout 1200 python -m pytest tests -m gpu -q -n 2 > gpurun_out/r06_gpu_tests.txt 2>&1; tail -15 gpurun_out/r06_gpu_tests.txt
