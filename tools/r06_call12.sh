cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -n 2 > gpurun_out/r06_gpu_tests.txt 2>&1; tail -4 gpurun_out/r06_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
