#!/bin/bash
# GPU box: A/B of shortest_path(T) (C3 graph, best of 30, host clock) against tools/bin/libwfst_amd_prev.so, phase stamps,
# the relaxation soak and the relaxation parity tests.   usage: tools/r3_ab.sh <tag> [soak seconds] [pytest -k expression|none]
set -u
export TMPDIR=/tmp
TAG=${1:-ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for cfg in "default" "WFST_LIB_PATH=tools/bin/libwfst_amd_prev.so" ${EXTRA_CFGS:-}; do
  if [ "$cfg" = "default" ]; then e=""; else e="${cfg//+/ }"; fi
  echo "== $cfg" >> $OUT/timing.txt
  env $e timeout -k 5 120 python tools/sp_repeat.py 1000000 30 >> $OUT/timing.txt 2>&1
done
grep -v amdgpu.ids $OUT/timing.txt
WFST_SSSP_MBOX_TRACE=/tmp/mbox_trace.bin timeout -k 5 120 python tools/sp_repeat.py 1000000 4 > /dev/null 2>&1 && python tools/mbox_phases.py /tmp/mbox_trace.bin > $OUT/phases.txt 2>&1
timeout -k 5 200 python tools/soak_sssp.py ${2:-30} 30000 > $OUT/soak.txt 2>&1
tail -1 $OUT/soak.txt
K=${3:-mailbox or config3_benched}
if [ "$K" != "none" ]; then
  timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > $OUT/parity.txt 2>&1
  tail -2 $OUT/parity.txt
fi
