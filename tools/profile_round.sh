#!/bin/bash
# Produces the rocprofv3 evidence for bench.py's numbers (run on the GPU box via gpurun):
#   gpurun_out/prof_<tag>/   --kernel-trace --stats of the default bench command
#   gpurun_out/pmc_fetch_<tag>/, pmc_write_<tag>/   separate --pmc passes (HBM traffic)
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o s -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -o s -- python $R/bench.py --no-cpu-baseline --steps 200 --warmup 50 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -o s -- python $R/bench.py --no-cpu-baseline --steps 200 --warmup 50 > /dev/null 2>&1
ls $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_fetch_$TAG $R/gpurun_out/pmc_write_$TAG
tail -1 $R/gpurun_out/prof_$TAG.json | cut -c1-200
