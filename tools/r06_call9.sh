cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_two_ranks_one_gpu.py tests/test_gpu_parity.py -m gpu -q -x -k "two_ranks or oversubscrib or past_its_episode or captured_step_follows or reward_log or hipgraph or record_source" > gpurun_out/r06_new_tests.txt 2>&1; tail -30 gpurun_out/r06_new_tests.txt
