#!/bin/bash
# GPU box: parity of the relaxation kernels + timing / per-launch trace / phase stamps of shortest_path(T) (C3 graph)
set -u
export TMPDIR=/tmp
TAG=${1:-r3a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[t+$(( $(date +%s) - T0 ))s] $1"; }
for cfg in "default" "WFST_SSSP_NARROW=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
  if [ "$cfg" = "default" ]; then e=""; else e="$cfg"; fi
  echo "== $cfg" >> $OUT/timing.txt
  env $e timeout 300 python tools/sp_repeat.py 1000000 30 >> $OUT/timing.txt 2>&1
done
grep -v amdgpu.ids $OUT/timing.txt
stamp timing
WFST_SSSP_MBOX_TRACE=/tmp/mbox_trace.bin timeout 300 python tools/sp_repeat.py 1000000 4 > /dev/null 2>&1 && python tools/mbox_phases.py /tmp/mbox_trace.bin > $OUT/phases.txt 2>&1
stamp phases
timeout 300 python tools/dbg_big.py > $OUT/dbg_big.txt 2>&1
grep -v amdgpu.ids $OUT/dbg_big.txt
stamp dbg_big
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mailbox or chasing or shortest_distance_matches or single_shortest" > $OUT/parity.txt 2>&1
tail -3 $OUT/parity.txt
stamp parity
timeout 200 python tools/soak_sssp.py 90 20000 > $OUT/soak.txt 2>&1
tail -2 $OUT/soak.txt
stamp soak
timeout 300 python tools/sweep_compare.py 1000000 1 > $OUT/sweeps.txt 2>&1
stamp sweeps
