cd $GRAFT_REPO_ROOT
SQ=1 bash tools/profile_round6.sh P2 C4 C2@65536 > gpurun_out/r06_profile_c.txt 2>&1
tail -6 gpurun_out/r06_profile_c.txt | cut -c1-200
ls gpurun_out/summ | grep -i "p2\|c4_\|65536"
