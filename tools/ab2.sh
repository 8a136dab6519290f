#!/bin/bash
# Development: like tools/ab_variants.sh with a per-run timeout, --no-workloads and extra bench arguments.
#   tools/ab2.sh "<bench args>" <rounds> name1 name2 ...      (libraries tools/bin/libaie_<name>.so)
ARGS=$1; ROUNDS=$2; shift 2
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    AIE_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/libaie_$v.so timeout 120 python bench.py $ARGS --no-cpu-baseline --no-workloads 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('$v', '%.4f ms/launch' % r['avg_launch_ms'], '%.1f M agent-steps/s' % (d['value']/1e6), flush=True)
except Exception as e:
    print('$v', 'failed', e, flush=True)"
  done
done
