cd $GRAFT_REPO_ROOT
bash tools/ab2.sh "--workload C2@16384 --steps 200 --warmup 20" 3 rt ntt
