#!/bin/bash
# Round-3 rocprofv3 evidence for bench.py's numbers (run on the GPU box via gpurun), per workload:
#   gpurun_out/prof_r03_<wl>/                 --kernel-trace --stats of `bench.py --workload <wl> --no-workloads`
#   gpurun_out/pmc_r03_<wl>_{FETCH,WRITE}_SIZE/   separate --pmc passes (HBM traffic; never combined with other traces)
#   gpurun_out/r03_<wl>_sq_counters.json      SQ instruction / cycle counters (tools/sq_passes.sh)
# and the summaries that go to profiles/: gpurun_out/summ/r03_<wl>_{bench.json,kernel_stats.csv,pmc.json,sq_counters.json}
# usage: tools/profile_round3.sh C2 [C3 C4 C5 ...]
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/summ
cd /tmp && export TMPDIR=/tmp
for WL in "$@"; do
  case $WL in C5) STEPS=100; WARM=10;; C4*) STEPS=300; WARM=30;; *) STEPS=600; WARM=50;; esac
  case $WL in C5) KN=aie_ose_step_kernel;; C4*) KN=aie_covid_step_kernel;; C2p) KN=aie_jit_step;; *) KN=aie_step_kernel;; esac
  w=$(echo $WL | tr 'A-Z' 'a-z')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_$WL -o s -- \
    python $R/bench.py --workload $WL --no-workloads --no-cpu-baseline --steps $STEPS --warmup $WARM > $R/gpurun_out/prof_r03_$WL.json 2> $R/gpurun_out/prof_r03_$WL.err
  tail -1 $R/gpurun_out/prof_r03_$WL.json > $R/gpurun_out/summ/r03_${w}_bench.json
  find $R/gpurun_out/prof_r03_$WL -name "*kernel_stats.csv" | head -1 | xargs -r head -12 > $R/gpurun_out/summ/r03_${w}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_r03_${WL}_$C -o s -- \
      python $R/bench.py --workload $WL --no-workloads --no-cpu-baseline --steps $((STEPS / 3)) --warmup $WARM > /dev/null 2> $R/gpurun_out/pmc_r03_${WL}_$C.err
  done
  python3 $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_r03_${WL}_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
    $(find $R/gpurun_out/pmc_r03_${WL}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/summ/r03_${w}_pmc.json $KN > /dev/null 2> $R/gpurun_out/summ/pmc_$w.err
  $R/tools/sq_passes.sh $WL --no-workloads > $R/gpurun_out/r03_sq_$WL.txt 2>&1
  cp $R/gpurun_out/r03_${WL}_sq_counters.json $R/gpurun_out/summ/r03_${w}_sq_counters.json 2>/dev/null
  head -3 $R/gpurun_out/summ/r03_${w}_kernel_stats.csv | cut -c1-160
  rm -rf $R/gpurun_out/prof_r03_$WL $R/gpurun_out/pmc_r03_${WL}_* $R/gpurun_out/sq_${WL}_*
done
