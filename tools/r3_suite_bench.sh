#!/bin/bash
# GPU box: the whole -m gpu suite, then the default bench line
set -u
export TMPDIR=/tmp
TAG=${1:-s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite.txt 2>&1
tail -4 $OUT/gpu_suite.txt
echo "[t+$(( $(date +%s) - T0 ))s] suite"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json
echo "[t+$(( $(date +%s) - T0 ))s] bench"
