#!/bin/bash
# GPU box: 8192-state blocks — parity test, then timings at 2M states (resident log 13 vs one launch per level vs atomic)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4i}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "8192_state_blocks or mailbox_sweeps_do_not_change or config3_benched or beyond_2_20" > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
timeout 600 python tools/r4_quick.py 2000000 12 "resident13:" "one_level:WFST_SSSP_RESIDENT=0" "atomic:WFST_SSSP_MAILBOX=0" > $OUT/timing2m.txt 2>&1
grep -v amdgpu.ids $OUT/timing2m.txt
timeout 600 python tools/r4_quick.py 1000000 12 "resident12:" > $OUT/timing1m.txt 2>&1
grep -v amdgpu.ids $OUT/timing1m.txt
