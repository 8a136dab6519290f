cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
V=$PWD/ai-economist_amd/csrc/variants
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c5/gputests_v5.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', '%.4f ms/launch' % r['avg_launch_ms'], '%.1f M agent-steps/s' % (d['value']/1e6), 'resets %d %.2f ms' % (r['reset_launches_in_region'], r['reset_ms_in_region']), d['config']['kernel_specialisation'])" >> gpurun_out/c5/ab.txt; }
for WL in C2 C3 C1; do
  for v in v4 v5 v4 v5; do
    AIE_HIP_LIBRARY=$V/libaie_$v.so timeout 200 python bench.py --workload $WL --no-cpu-baseline --no-workloads --steps 400 --warmup 50 2>/dev/null | line "$WL $v"
  done
done
timeout 200 python bench.py --workload C2p --no-cpu-baseline --no-workloads --steps 400 --warmup 50 2>/dev/null | line "C2p jit"
timeout 200 python bench.py --workload C2 --generic-kernel --no-cpu-baseline --no-workloads --steps 400 --warmup 50 2>/dev/null | line "C2 generic v5"
AIE_HIP_LIBRARY=$V/libaie_v2.so timeout 200 python bench.py --workload C2 --generic-kernel --no-cpu-baseline --no-workloads --steps 400 --warmup 50 2>/dev/null | line "C2 generic v2"
timeout 200 python bench.py --workload C3 --generic-kernel --no-cpu-baseline --no-workloads --steps 400 --warmup 50 2>/dev/null | line "C3 generic v5"
AIE_HIP_LIBRARY=$V/libaie_v2.so timeout 200 python bench.py --workload C3 --generic-kernel --no-cpu-baseline --no-workloads --steps 400 --warmup 50 2>/dev/null | line "C3 generic v2"
timeout 120 python tools/block_trace.py 10 b2b > gpurun_out/c5/trace10.txt 2>&1
grep -n 'passed\|failed' gpurun_out/c5/gputests_v5.txt; cat gpurun_out/c5/ab.txt
