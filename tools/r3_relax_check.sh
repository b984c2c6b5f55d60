#!/bin/bash
# GPU box: parity of the relaxation kernels + timing / per-launch trace / phase stamps of shortest_path(T) (C3 graph)
set -u
export TMPDIR=/tmp
TAG=${1:-r3a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mailbox or chasing or shortest_distance_matches or single_shortest" > $OUT/parity.txt 2>&1
tail -5 $OUT/parity.txt
for cfg in "default" "WFST_SSSP_NARROW=0" "WFST_SSSP_HINT=0" "WFST_SSSP_HINT=1" "WFST_SSSP_NARROW=0 WFST_SSSP_HINT=0"; do
  if [ "$cfg" = "default" ]; then e=""; else e="$cfg"; fi
  echo "== $cfg" >> $OUT/timing.txt
  env $e timeout 300 python tools/sp_repeat.py 1000000 30 >> $OUT/timing.txt 2>&1
done
cat $OUT/timing.txt
timeout 300 python tools/sweep_compare.py 1000000 1 > $OUT/sweeps.txt 2>&1
WFST_SSSP_MBOX_TRACE=/tmp/mbox_trace.bin timeout 300 python tools/sp_repeat.py 1000000 4 > /dev/null 2>&1 && python tools/mbox_phases.py /tmp/mbox_trace.bin > $OUT/phases.txt 2>&1
timeout 240 python tools/soak_sssp.py 120 5000 > $OUT/soak.txt 2>&1
tail -2 $OUT/soak.txt
