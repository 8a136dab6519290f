cd $GRAFT_REPO_ROOT
SQ=1 C2PI=1 bash tools/profile_round6.sh C2 C3 C1 C1f C2@16384 C2f C3f C4x C4xu C5 P2 C4 C2@65536 C2v > gpurun_out/r06_profile_all.txt 2>&1
tail -5 gpurun_out/r06_profile_all.txt | cut -c1-200
ls gpurun_out/summ | wc -l
