#!/bin/bash
# GPU box, round 4, first contact of the resident kernel: parity subset, A/B timing, level trace
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4a}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mailbox_sweeps_do_not_change or config3_benched" > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 600 python tools/r4_quick.py 1000000 20 "resident:" "one_level:WFST_SSSP_RESIDENT=0" "atomic:WFST_SSSP_MAILBOX=0" > $OUT/timing.txt 2>&1
grep -v amdgpu.ids $OUT/timing.txt
WFST_SSSP_RES_TRACE=/tmp/res_trace.bin timeout 300 python tools/sp_repeat.py 1000000 4 > /dev/null 2>&1 && python tools/res_levels.py /tmp/res_trace.bin > $OUT/levels.txt 2>&1
cat $OUT/levels.txt
