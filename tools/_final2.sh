cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final2
timeout 1200 bash tools/profile_round3.sh C4 C4x > gpurun_out/final2/profile.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final2/bench_default.json 2> gpurun_out/final2/bench_default.err
timeout 600 python bench.py --gpus 1 > gpurun_out/final2/bench_long.json 2> gpurun_out/final2/bench_long.err
cat gpurun_out/final2/profile.txt | cut -c1-180; tail -c 300 gpurun_out/final2/bench_default.json
