#!/bin/bash
# Runs on the GPU box (via gpurun): the round-2 evidence set.  usage: tools/profile_round2.sh <tag>
#   bench line (default command), rocprofv3 kernel trace of the bench, PMC passes of the relaxation kernel and of the
#   fused batch kernel, mailbox phase stamps, a kernel trace of one large composition on the wide driver, the KDELTA table
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --kernel-trace -d $OUT -o sp_alone -- python $R/tools/sp_repeat.py 1000000 12 > $OUT/sp_alone.log 2>&1
rocprofv3 --kernel-trace -d $OUT -o wide -- python $R/tools/lookahead_timing.py 10000,100,3,16,16 > $OUT/wide.log 2>&1
cd $R
WFST_SSSP_MBOX_TRACE=/tmp/mb.bin python tools/sp_repeat.py 1000000 5 > /dev/null 2>&1; python tools/mbox_phases.py /tmp/mb.bin > $OUT/mbox_phases.txt
python tools/sweep_compare.py 1000000 0,1 > $OUT/sweep_compare.txt 2>&1
python tools/lookahead_timing.py 300,20,3,8,8 2000,50,3,12,12 10000,100,3,16,16 40000,100,3,16,16 2>&1 | grep "^ *[0-9n]" > $OUT/wide_timing.txt
python tools/step_breakdown.py 2>&1 | grep -v amdgpu.ids > $OUT/step_breakdown.txt
if [ -z "${SKIP_SLOW:-}" ]; then  # (two minutes of CPU oracle between them)
  python tools/kdelta_gap.py > $OUT/kdelta_gap.txt 2>&1
  python tools/rm_epsilon_timing.py > $OUT/rm_epsilon_timing.txt 2>&1
fi
tools/pmc_relax.sh ${TAG}_mbox 1 > $OUT/pmc_mbox.txt 2>&1
tools/pmc_batch.sh ${TAG}_string > $OUT/pmc_string.txt 2>&1
ls $OUT
