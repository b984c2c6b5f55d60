#!/bin/bash
# GPU box: A/B of shortest_path(T) on the C3 graph + level trace of the resident launch. usage: r4_ab.sh TAG [cfg ...]
set -u
export TMPDIR=/tmp
TAG=${1:-r4x}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ $# -eq 0 ]; then set -- "resident:" "one_level:WFST_SSSP_RESIDENT=0"; fi
timeout 600 python tools/r4_quick.py ${R4_STATES:-1000000} ${R4_REPS:-20} "$@" > $OUT/timing.txt 2>&1
grep -v amdgpu.ids $OUT/timing.txt
WFST_SSSP_RES_TRACE=/tmp/res_trace.bin timeout 300 python tools/sp_repeat.py ${R4_STATES:-1000000} 8 > /dev/null 2>&1 && python tools/res_levels.py /tmp/res_trace.bin > $OUT/levels.txt 2>&1
cat $OUT/levels.txt
