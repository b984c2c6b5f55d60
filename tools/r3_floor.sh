#!/bin/bash
# GPU box: tools/ubench_floor under rocprofv3 (kernel durations per variant) and alone (launch to launch)
set -u
export TMPDIR=/tmp
TAG=${1:-floor}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
R=$PWD
timeout -k 5 100 $R/tools/bin/ubench_floor > $OUT/host.txt 2>&1
cat $OUT/host.txt
cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT -o floor -f csv -- $R/tools/bin/ubench_floor > $OUT/prof.log 2>&1
ls $OUT
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70].ljust(72), r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"])
PY
