cd $GRAFT_REPO_ROOT
SQ=1 C2PI=1 bash tools/profile_round6.sh C2 C3 C1 C1f C2@16384 C2f C3f > gpurun_out/r06_profile_a.txt 2>&1
tail -30 gpurun_out/r06_profile_a.txt | cut -c1-200
ls gpurun_out/summ | wc -l
