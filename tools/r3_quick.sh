#!/bin/bash
# GPU box: quick A/B of shortest_path(T) on the C3 graph (best of 30, host clock) + phase stamps
set -u
export TMPDIR=/tmp
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for cfg in "default" "WFST_SSSP_NARROW=0" ${EXTRA_CFGS:-}; do
  if [ "$cfg" = "default" ]; then e=""; else e="${cfg//+/ }"; fi
  echo "== $cfg" >> $OUT/timing.txt
  env $e timeout 300 python tools/sp_repeat.py 1000000 30 >> $OUT/timing.txt 2>&1
done
grep -v amdgpu.ids $OUT/timing.txt
WFST_SSSP_MBOX_TRACE=/tmp/mbox_trace.bin timeout 300 python tools/sp_repeat.py 1000000 4 > /dev/null 2>&1 && python tools/mbox_phases.py /tmp/mbox_trace.bin > $OUT/phases.txt 2>&1
timeout 200 python tools/soak_sssp.py ${SOAK_S:-30} 30000 > $OUT/soak.txt 2>&1
tail -1 $OUT/soak.txt
for sz in ${BIG_SIZES:-}; do
  for m in 1 0; do
    echo "== states $sz WFST_SSSP_MAILBOX=$m" >> $OUT/big.txt
    WFST_SSSP_MAILBOX=$m timeout -k 5 200 python tools/sp_repeat.py $sz 10 >> $OUT/big.txt 2>&1
  done
done
[ -f $OUT/big.txt ] && grep -v amdgpu.ids $OUT/big.txt
