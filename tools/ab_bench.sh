#!/bin/bash
# Development: A/B of two builds of the library on ONE box (launch times move by +-6 % between boxes):
#   tools/ab_bench.sh <workload> <baseline .so> [steps] [rounds]   -> alternating runs, ms per launch of each
WL=$1; BASE=$2; STEPS=${3:-200}; ROUNDS=${4:-3}
for r in $(seq $ROUNDS); do
  for which in base new; do
    if [ $which = base ]; then export AIE_HIP_LIBRARY=$BASE; else unset AIE_HIP_LIBRARY; fi
    python bench.py --workload $WL --no-cpu-baseline --steps $STEPS --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$which', '%.4f ms/launch' % r['avg_launch_ms'], '%.1f M agent-steps/s' % (d['value']/1e6), 'resets %d %.2f ms' % (r['reset_launches_in_region'], r['reset_ms_in_region']))"
  done
done
