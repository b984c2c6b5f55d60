#!/bin/bash
# GPU box: the tests added in round 3 + look-ahead create timing at configs[4] size
set -u
export TMPDIR=/tmp
TAG=${1:-t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
T0=$(date +%s)
timeout -k 5 420 python -m pytest tests -x -q -m gpu -k "reference_tie_order or nshortest or rccl_single or k2 or shortest_path" > $OUT/new_tests.txt 2>&1
tail -15 $OUT/new_tests.txt
echo "[t+$(( $(date +%s) - T0 ))s] tests"
true
grep -v amdgpu $OUT/config5.txt | tail -12
echo "[t+$(( $(date +%s) - T0 ))s] config5"
