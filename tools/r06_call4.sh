cd $GRAFT_REPO_ROOT
for v in base v2; do echo $v; AIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/libaie_$v.so timeout 200 python tools/step_timing.py 4 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
for v in base v2; do
AIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/libaie_$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$v -o s -- python $GRAFT_REPO_ROOT/tools/step_timing.py 4 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -r head -5 | cut -c1-200
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$v
done
