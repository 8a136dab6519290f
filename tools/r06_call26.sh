cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final_driver_window.txt 2> gpurun_out/r06_final_driver_window.err
tail -1 gpurun_out/r06_final_driver_window.txt > gpurun_out/r06_final_bench.json
cp bench_detail.json gpurun_out/r06_final_bench_detail.json
python bench.py --gpus 1 --no-workloads --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_c2_long_window.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-workloads --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_c2_driver_window_2.json
cut -c1-1500 gpurun_out/r06_final_bench.json
echo
cut -c1-300 gpurun_out/r06_c2_long_window.json
echo
cut -c1-300 gpurun_out/r06_c2_driver_window_2.json
