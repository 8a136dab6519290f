cd $GRAFT_REPO_ROOT
SQ=1 C2PI=1 bash tools/profile_round6.sh C2 C3 C1 C1f C2@16384 C2f C3f C4x C4xu C5 P2 C4 C2@65536 C2v > gpurun_out/r06_profile_all.txt 2>&1
for w in c1 c1f c2 c2_e16384 c2_e65536 c2f c2v c3 c3f c4 c4x c4xu c5 p2; do for k in bench.json kernel_stats.csv pmc.json sq_counters.json; do cp gpurun_out/summ/r06_${w}_$k profiles/; done; done
cp gpurun_out/summ/r06_c2pi_bench.json gpurun_out/summ/r06_c2pi_kernel_stats.csv profiles/
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final_driver_window.txt 2> gpurun_out/r06_final_driver_window.err
tail -1 gpurun_out/r06_final_driver_window.txt > gpurun_out/r06_final_bench.json
cp bench_detail.json gpurun_out/r06_final_bench_detail.json
python bench.py --gpus 1 --no-workloads --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_c2_long_window.json
cut -c1-200 gpurun_out/r06_final_bench.json
echo
cut -c1-200 gpurun_out/r06_c2_long_window.json
echo
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed" > gpurun_out/r06_final_gpu_tests.txt
cat gpurun_out/r06_final_gpu_tests.txt
