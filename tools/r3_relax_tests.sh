#!/bin/bash
set -u
export TMPDIR=/tmp
TAG=${1:-rt}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 5 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mailbox or chasing or shortest_distance_matches or single_shortest or config3_benched or chain_timing or nshortest" > $OUT/parity.txt 2>&1
tail -3 $OUT/parity.txt
SOAK_S=60 bash tools/r3_quick.sh $TAG
python - <<'PY'
import sys
sys.path.insert(0, "tools")
PY
