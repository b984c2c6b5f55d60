#!/bin/bash
# Runs on the GPU box (via gpurun): the round-3 evidence set.  usage: tools/profile_round3.sh <tag>
#   the whole -m gpu suite, the default bench line, rocprofv3 kernel traces (bench, the relaxation alone, a 5M-state solve
#   under both relaxation kernels), mailbox phase stamps, the per-launch comparison, PMC passes of the relaxation kernel
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $1"; }
timeout -k 5 600 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite.txt 2>&1
tail -3 $OUT/gpu_suite.txt
stamp suite
timeout -k 5 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
tail -c 600 $OUT/bench_line.json; tail -3 $OUT/bench_line.err
stamp bench
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT -o sp_alone -- python $R/tools/sp_repeat.py 1000000 12 > $OUT/sp_alone.log 2>&1
stamp traces
for m in 1 0; do
  WFST_SSSP_MAILBOX=$m timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT -o sp5m_mode$m -- python $R/tools/sp_repeat.py 5000000 6 > $OUT/sp5m_mode$m.log 2>&1
done
stamp sp5m
cd $R
WFST_SSSP_MBOX_TRACE=/tmp/mb.bin timeout -k 5 200 python tools/sp_repeat.py 1000000 5 > /dev/null 2>&1; python tools/mbox_phases.py /tmp/mb.bin > $OUT/mbox_phases.txt
timeout -k 5 300 python tools/sweep_compare.py 1000000 0,1 > $OUT/sweep_compare.txt 2>&1
stamp phases
timeout -k 5 600 tools/pmc_relax.sh ${TAG}_mbox 1 > $OUT/pmc_mbox.txt 2>&1
stamp pmc
ls $OUT
