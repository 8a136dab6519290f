cd $GRAFT_REPO_ROOT
bash tools/ab2.sh "--steps 2000 --warmup 200" 3 base tl
