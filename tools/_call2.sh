cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
V=$PWD/ai-economist_amd/csrc/variants
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c2/gputests_v2.txt
if ! grep -q ' passed' gpurun_out/c2/gputests_v2.txt || grep -q 'failed' gpurun_out/c2/gputests_v2.txt; then
  ( AIE_HIP_LIBRARY=$V/libaie_v1.so timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c2/gputests_v1.txt
fi
for r in 1 2; do
for WL in C2 C3 C1; do
  for v in v0 v1 v2; do
    AIE_HIP_LIBRARY=$V/libaie_$v.so timeout 200 python bench.py --workload $WL --no-cpu-baseline --no-workloads --steps 600 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$WL $v', '%.4f ms/launch' % r['avg_launch_ms'], '%.1f M agent-steps/s' % (d['value']/1e6), 'resets %d %.2f ms' % (r['reset_launches_in_region'], r['reset_ms_in_region']))" >> gpurun_out/c2/ab.txt
  done
done
done
timeout 300 python tools/spec_ablate.py 4 10 > gpurun_out/c2/ablate_v2.txt 2>&1
cat gpurun_out/c2/gputests_v2.txt | tail -4; cat gpurun_out/c2/ab.txt
