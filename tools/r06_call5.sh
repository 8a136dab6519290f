cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in "$@"; do echo -n "$v: "; AIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/libaie_$v.so timeout 200 python tools/step_timing.py 4 2>&1 | tail -1; done
done
