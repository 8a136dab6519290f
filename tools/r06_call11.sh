cd $GRAFT_REPO_ROOT
for v in v6 v7nt v7; do echo -n "$v: "; AIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/libaie_$v.so timeout 200 python tools/step_timing.py 4 2>&1 | tail -1; done
bash tools/ab2.sh "--workload C2 --steps 2000 --warmup 100" 2 v6 v7nt v7 2>&1
bash tools/ab2.sh "--workload C2 --envs-per-gpu 16384 --steps 600 --warmup 50" 1 v6 v7nt v7 2>&1
bash tools/ab2.sh "--workload C3 --steps 1000 --warmup 100" 1 v6 v7nt v7 2>&1
bash tools/ab2.sh "--workload C1 --steps 1000 --warmup 100" 1 v6 v7nt v7 2>&1
