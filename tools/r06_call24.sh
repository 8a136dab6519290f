cd $GRAFT_REPO_ROOT
SQ=1 bash tools/profile_round6.sh C1f C4x C4xu C5 P2 C4 C2@65536 C2v > gpurun_out/r06_profile_b.txt 2>&1
tail -12 gpurun_out/r06_profile_b.txt | cut -c1-200
ls gpurun_out/summ | wc -l
