#!/bin/bash
# Runs on the GPU box (via gpurun): the end-of-round evidence set.  usage: tools/profile_round.sh <tag>
#   kernel traces (bench step; the relaxation alone at 1M with the chain ALSO timed by HIP events in the same process; 5M;
#   the wide look-ahead driver), level stamps of the resident launch, PMC passes of the relaxation at 1M (separate passes).
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
OUTP=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT $OUTP
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $1"; }
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT -o sp_alone -- python $R/tools/sp_repeat.py 1000000 12 > $OUT/sp_alone.log 2>&1
timeout -k 5 300 python $R/tools/sp_repeat.py 1000000 12 > $OUT/sp_alone_unprofiled.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT -o sp_2m -- python $R/tools/sp_repeat.py 2000000 8 > $OUT/sp_2m.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT -o sp_5m -- python $R/tools/sp_repeat.py 5000000 6 > $OUT/sp_5m.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT -o wide -- python $R/tools/wide_lookahead_run.py 5000000 4 > $OUT/wide.log 2>&1
stamp traces
cd $R
WFST_SSSP_RES_TRACE=/tmp/res.bin timeout -k 5 200 python tools/sp_repeat.py 1000000 8 > /dev/null 2>&1; python tools/res_levels.py /tmp/res.bin > $OUT/res_levels.txt
stamp levels
cd /tmp
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $OUTP -o $name -- python $R/tools/sp_repeat.py 1000000 12 > $OUTP/$name.log 2>&1 || echo "pass $name failed: $(tail -2 $OUTP/$name.log)"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
stamp pmc
cd $R
for db in trace sp_alone sp_2m sp_5m wide; do
  f=$(ls $OUT/${db}_results.db 2>/dev/null | head -1)
  [ -n "$f" ] && { echo "## $db"; python tools/rocpd_summary.py trace $f; } >> $OUT/kernel_tables.md 2>&1
done
python tools/pmc_summary.py $OUTP > $OUT/pmc_summary.txt 2>&1
grep -h "chain by HIP events\|best of" $OUT/sp_alone.log $OUT/sp_alone_unprofiled.log
ls $OUT | head -40
