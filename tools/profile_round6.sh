#!/bin/bash
# Round-6 rocprofv3 evidence for bench.py's numbers (run on the GPU box via gpurun), per workload:
#   --kernel-trace --stats of `bench.py --workload <wl> --no-workloads`, separate FETCH_SIZE / WRITE_SIZE --pmc passes
#   (HBM traffic; never combined with other traces) -> summaries in
#   gpurun_out/summ/r06_<wl>_{bench.json,kernel_stats.csv,pmc.json} (copied to profiles/ by hand).
#   SQ=1 in the environment adds the SQ instruction / cycle passes (tools/sq_passes.sh).
# usage: tools/profile_round6.sh C2 C2@16384 C2@65536 C4xu C2f ...
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/summ
cd /tmp && export TMPDIR=/tmp
for WL in "$@"; do
  case $WL in C5) STEPS=100; WARM=20;; C4*) STEPS=300; WARM=30;; C2@65536) STEPS=90; WARM=10;; C2@16384) STEPS=300; WARM=30;; *) STEPS=600; WARM=50;; esac
  case $WL in C5) KN=aie_ose_step_kernel;; C4x*) KN=aie_covid_step_kernel+aie_covid_window_kernel;; C4) KN=aie_covid_step_kernel;; *) KN=aie_step_kernel;; esac
  w=$(echo $WL | tr 'A-Z' 'a-z' | sed 's/@/_e/')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06_$w -o s -- \
    python $R/bench.py --workload $WL --no-workloads --no-cpu-baseline --steps $STEPS --warmup $WARM > $R/gpurun_out/prof_r06_$w.json 2> $R/gpurun_out/prof_r06_$w.err
  tail -1 $R/gpurun_out/prof_r06_$w.json > $R/gpurun_out/summ/r06_${w}_bench.json  # (the compact driver line; the full one is the line before it)
  find $R/gpurun_out/prof_r06_$w -name "*kernel_stats.csv" | head -1 | xargs -r head -12 > $R/gpurun_out/summ/r06_${w}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_r06_${w}_$C -o s -- \
      python $R/bench.py --workload $WL --no-workloads --no-cpu-baseline --steps $((STEPS / 3)) --warmup $WARM > /dev/null 2> $R/gpurun_out/pmc_r06_${w}_$C.err
  done
  python3 $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_r06_${w}_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
    $(find $R/gpurun_out/pmc_r06_${w}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/summ/r06_${w}_pmc.json $KN > /dev/null 2> $R/gpurun_out/summ/pmc_$w.err
  if [ -n "$SQ" ]; then
    $R/tools/sq_passes.sh $WL --no-workloads > $R/gpurun_out/r06_sq_$w.txt 2>&1
    cp $R/gpurun_out/r04_${WL}_sq_counters.json $R/gpurun_out/summ/r06_${w}_sq_counters.json 2>/dev/null
  fi
  head -4 $R/gpurun_out/summ/r06_${w}_kernel_stats.csv | cut -c1-160
  rm -rf $R/gpurun_out/prof_r06_$w $R/gpurun_out/pmc_r06_${w}_* $R/gpurun_out/sq_${WL}_*
done
# the policy-in-the-loop workload: kernel stats only (which launches a replayed iteration consists of)
if [ -n "$C2PI" ]; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06_c2pi -o s -- \
    python $R/bench.py --workload C2pi --steps 300 --warmup 30 > $R/gpurun_out/summ/r06_c2pi_bench.json 2> $R/gpurun_out/prof_r06_c2pi.err
  find $R/gpurun_out/prof_r06_c2pi -name "*kernel_stats.csv" | head -1 | xargs -r head -24 > $R/gpurun_out/summ/r06_c2pi_kernel_stats.csv
  head -12 $R/gpurun_out/summ/r06_c2pi_kernel_stats.csv | cut -c1-150
  rm -rf $R/gpurun_out/prof_r06_c2pi
fi
