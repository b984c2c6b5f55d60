#!/bin/bash
# GPU box: the end-of-round check in one call — the whole -m gpu suite, then the evidence set of tools/profile_round2.sh
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/final/gpu_suite.txt 2>&1
tail -3 gpurun_out/final/gpu_suite.txt
SKIP_SLOW=1 timeout 900 bash tools/profile_round2.sh ${1:-r02m} > gpurun_out/final/profile.log 2>&1
tail -5 gpurun_out/final/profile.log
