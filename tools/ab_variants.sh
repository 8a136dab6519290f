#!/bin/bash
# Development: launch times of several builds of the library (tools/bin/libaie_<name>.so) on ONE box, alternating.
#   tools/ab_variants.sh <workload> <steps> <rounds> name1 name2 ...
WL=$1; STEPS=$2; ROUNDS=$3; shift 3
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    AIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/libaie_$v.so python bench.py --workload $WL --no-cpu-baseline --steps $STEPS --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', '%.4f ms/launch' % r['avg_launch_ms'], '%.1f M agent-steps/s' % (d['value']/1e6))"
  done
done
