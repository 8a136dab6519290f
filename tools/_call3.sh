cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
V=$PWD/ai-economist_amd/csrc/variants
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c3/gputests_v3.txt
( AIE_HIP_LIBRARY=$V/libaie_v3e.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c3/gputests_v3e.txt
for WL in C2 C3; do
  for v in v2 v3 v3a v3b v3c v3d v3e v3f v2 v3; do
    AIE_HIP_LIBRARY=$V/libaie_$v.so timeout 200 python bench.py --workload $WL --no-cpu-baseline --no-workloads --steps 400 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$WL $v', '%.4f ms/launch' % r['avg_launch_ms'], '%.1f M agent-steps/s' % (d['value']/1e6), 'resets %d %.2f ms' % (r['reset_launches_in_region'], r['reset_ms_in_region']))" >> gpurun_out/c3/ab.txt
  done
done
tail -3 gpurun_out/c3/gputests_v3.txt; tail -3 gpurun_out/c3/gputests_v3e.txt; cat gpurun_out/c3/ab.txt
