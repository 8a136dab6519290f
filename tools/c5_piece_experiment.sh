#!/bin/bash
# Development (round 5): how much of the one-step-economy launch time on THIS box is the arena's piece size?  Fresh
# processes with forced 16 / 64 / 128 MiB pieces and the default.  Output per process: label, piece used, avg_launch_ms.
R=$GRAFT_REPO_ROOT
run() {
  python $R/bench.py --workload C5 --steps 60 --warmup 20 --no-cpu-baseline --no-workloads --detail-file /dev/null 2>/dev/null | tail -2 | head -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', r.get('arena_piece_mib'), round(r['avg_launch_ms'],4))"
}
for P in 16 64 128; do AIE_ARENA_PIECE_MB=$P run forced$P; done
run default
