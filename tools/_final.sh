cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/final/gputests.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
timeout 1200 bash tools/profile_round3.sh C2 C3 C1 C5 > gpurun_out/final/profile.txt 2>&1
timeout 100 python tools/block_trace.py 4 b2b > gpurun_out/final/trace4.txt 2>&1
timeout 100 python tools/block_trace.py 10 b2b > gpurun_out/final/trace10.txt 2>&1
timeout 200 python tools/spec_ablate.py 4 10 > gpurun_out/final/ablate.txt 2>&1
grep -n 'passed\|failed' gpurun_out/final/gputests.txt; tail -c 600 gpurun_out/final/bench_default.json; cat gpurun_out/final/profile.txt | tail -20
