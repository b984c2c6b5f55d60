#!/bin/bash
# GPU box: parity subset for the relaxation kernels, then A/B timing + level trace. usage: r4_check.sh TAG [cfg ...]
set -u
export TMPDIR=/tmp
TAG=${1:-r4x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mailbox_sweeps_do_not_change or config3_benched" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
bash tools/r4_ab.sh "$@"
