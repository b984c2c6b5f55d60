#!/bin/bash
# GPU box: PMC passes (one counter group per pass, --kernel-trace only) over un-profiled shortest_path(T) solves of any size.
# usage: tools/pmc_any.sh <tag> <states> <reps> [VAR=val ...]      -> gpurun_out/pmc_<tag>/, summary on stdout
set -u
TAG=$1; STATES=$2; REPS=$3; shift 3
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
pass() {
  local name=$1; shift
  env "${ENVV[@]}" rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $name -- python $R/tools/sp_repeat.py $STATES $REPS > $OUT/$name.log 2>&1 || echo "pass $name failed: $(tail -2 $OUT/$name.log)"
}
ENVV=("$@" "PMC_DUMMY=1")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
cd $R
PMC_KERNELS=all python tools/pmc_summary.py $OUT
