cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
SQ=1 C2PI=1 bash tools/profile_round6.sh C2 C3 C1 C1f C2@16384 C2f C3f C4x C4xu C5 P2 C4 C2@65536 C2v > gpurun_out/r06_profile_all.txt 2>&1
tail -3 gpurun_out/r06_profile_all.txt | cut -c1-200
