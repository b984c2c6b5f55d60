#!/bin/bash
# GPU box: randomised differential soaks of every component against the oracle + the destroy-order test
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-soaks}
mkdir -p $OUT
timeout -k 5 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "handles_outlive or wide_driver or lookahead_compose_batch" > $OUT/t.txt 2>&1; tail -2 $OUT/t.txt
timeout -k 5 100 python tools/soak.py 60 40000 > $OUT/soak.txt 2>&1; tail -1 $OUT/soak.txt
timeout -k 5 100 python tools/soak_lookahead.py 60 40000 > $OUT/soak_la.txt 2>&1; tail -1 $OUT/soak_la.txt
timeout -k 5 80 python tools/soak_ops.py 40 40000 > $OUT/soak_ops.txt 2>&1; tail -1 $OUT/soak_ops.txt
WFST_COMPOSE_PATH=wide timeout -k 5 60 python tools/soak.py 30 50000 > $OUT/soak_wide.txt 2>&1; tail -1 $OUT/soak_wide.txt
