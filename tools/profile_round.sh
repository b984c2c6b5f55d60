#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of bench.py, then two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md §rocprofv3 PMC slots).
# usage: tools/profile_round.sh r01
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_write.log 2>&1
ls -la $OUT
