cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "policy_sampler or hipgraph" 2>&1 | tail -5
SKIPS=0,4,7,8 WAVES_LOG2=0,1,2 timeout 1000 python tools/sampler_timing.py 2>&1 | grep -E "skip|rror" | grep warm
