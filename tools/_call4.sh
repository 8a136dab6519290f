cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
V=$PWD/ai-economist_amd/csrc/variants
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c4/gputests_v4.txt
for WL in C2 C3 C1 C2p; do
  for v in v2 v3 v4n v4 v2 v4; do
    AIE_HIP_LIBRARY=$V/libaie_$v.so timeout 200 python bench.py --workload $WL --no-cpu-baseline --no-workloads --steps 400 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$WL $v', '%.4f ms/launch' % r['avg_launch_ms'], '%.1f M agent-steps/s' % (d['value']/1e6), 'resets %d %.2f ms' % (r['reset_launches_in_region'], r['reset_ms_in_region']))" >> gpurun_out/c4/ab.txt
  done
done
timeout 120 python tools/block_trace.py 4 b2b > gpurun_out/c4/trace4.txt 2>&1
timeout 120 python tools/block_trace.py 10 b2b > gpurun_out/c4/trace10.txt 2>&1
timeout 120 python tools/block_trace.py 4 generic b2b > gpurun_out/c4/trace4g.txt 2>&1
grep -n 'passed\|failed' gpurun_out/c4/gputests_v4.txt; cat gpurun_out/c4/ab.txt
