#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes over N un-profiled shortest_path(T) solves, one counter group per pass
# (MI355X_MICROARCH.md §rocprofv3 PMC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ has 8 slots, TCC 4).
# usage: tools/pmc_relax.sh <tag> <WFST_SSSP_MAILBOX value>
set -u
TAG=${1:-r02}
MODE=${2:-1}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
export WFST_SSSP_MAILBOX=$MODE
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $name -- python $R/tools/sp_repeat.py 1000000 6 > $OUT/$name.log 2>&1 || echo "pass $name failed: $(tail -2 $OUT/$name.log)"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass atom TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
cd $R
python tools/pmc_summary.py $OUT
