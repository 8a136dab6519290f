cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_rng_fast.py tests/test_gpu_parity.py -m gpu -q -x -k "fast or layout or reset or golden" --tb=short 2>&1 | tail -12 | cut -c1-300
