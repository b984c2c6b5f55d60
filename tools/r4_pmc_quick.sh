#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of the relaxation launches per solve (two separate PMC passes), for the env given as args
set -u
export TMPDIR=/tmp
R=$PWD
TAG=${1:-pq}; shift
OUTP=$R/gpurun_out/pmcq_$TAG
rm -rf $OUTP; mkdir -p $OUTP
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" rocprofv3 --pmc $c --kernel-trace -d $OUTP -o $c -- python $R/tools/sp_repeat.py 1000000 12 > $OUTP/$c.log 2>&1 || echo "pass $c failed"
done
cd $R
python tools/pmc_summary.py $OUTP | grep -A4 "resident_kernel\|traffic"
