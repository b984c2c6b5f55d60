#!/bin/bash
# GPU box: schedule-parameter sweeps of the relaxation (tools/param_sweep.py) on several graphs
set -u
export TMPDIR=/tmp
TAG=${1:-param}
mkdir -p gpurun_out/$TAG
for cfg in "1000000 12 10 fine" "1000000 10 4 fine" "1000000 10 20 fine" "2000000 8 10 fine" "300000 12 10 fine"; do
  timeout -k 5 300 python tools/param_sweep.py $cfg > gpurun_out/$TAG/sweep_${cfg// /_}.txt 2>&1
  grep -v amdgpu.ids gpurun_out/$TAG/sweep_${cfg// /_}.txt | head -14
done
