cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests.txt 2>&1; tail -5 gpurun_out/r06_gpu_tests.txt
bash tools/ab2.sh "--workload C2 --steps 2000 --warmup 100" 2 base v2 > gpurun_out/r06_ab_v2.txt 2>&1
echo "HIP_FORCE_DEV_KERNARG=1" >> gpurun_out/r06_ab_v2.txt
HIP_FORCE_DEV_KERNARG=1 bash tools/ab2.sh "--workload C2 --steps 2000 --warmup 100" 1 base v2 >> gpurun_out/r06_ab_v2.txt 2>&1
echo "HIP_FORCE_DEV_KERNARG=0" >> gpurun_out/r06_ab_v2.txt
HIP_FORCE_DEV_KERNARG=0 bash tools/ab2.sh "--workload C2 --steps 2000 --warmup 100" 1 base v2 >> gpurun_out/r06_ab_v2.txt 2>&1
cat gpurun_out/r06_ab_v2.txt
