#!/bin/bash
# GPU box: kernel trace of un-profiled shortest_path(T) solves (C3 graph) -> timeline of one solve
set -u
export TMPDIR=/tmp
TAG=${1:-tr}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace -d $OUT -o sp_alone -- python $R/tools/sp_repeat.py 1000000 12 > $OUT/sp_alone.log 2>&1
cd $R
grep "best of" $OUT/sp_alone.log
python - <<PY
import sys
sys.path.insert(0, "tools")
import rocpd_summary
rocpd_summary.trace("$OUT/sp_alone_results.db")
rocpd_summary.timeline("$OUT/sp_alone_results.db", -3)
PY
