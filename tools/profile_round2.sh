#!/bin/bash
# Round-2 rocprofv3 evidence for bench.py's numbers (run on the GPU box via gpurun), per workload:
#   gpurun_out/prof_r02_<wl>/   --kernel-trace --stats of `bench.py --workload <wl>`
#   gpurun_out/pmc_r02_<wl>_{fetch,write}/   separate --pmc passes (HBM traffic; never combined with other traces)
# usage: tools/profile_round2.sh C2 [C3 C4 C5]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for WL in "$@"; do
  case $WL in C5) STEPS=100; WARM=10;; C4) STEPS=300; WARM=30;; *) STEPS=600; WARM=50;; esac
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_$WL -o s -- \
    python $R/bench.py --workload $WL --no-cpu-baseline --steps $STEPS --warmup $WARM > $R/gpurun_out/prof_r02_$WL.json 2> $R/gpurun_out/prof_r02_$WL.err
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_r02_${WL}_$C -o s -- \
      python $R/bench.py --workload $WL --no-cpu-baseline --steps $((STEPS / 3)) --warmup $WARM > /dev/null 2> $R/gpurun_out/pmc_r02_${WL}_$C.err
  done
  find $R/gpurun_out/prof_r02_$WL -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-160
  tail -c 300 $R/gpurun_out/prof_r02_$WL.json
done
