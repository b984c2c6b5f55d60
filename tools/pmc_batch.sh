#!/bin/bash
# PMC passes over the fused batch kernel (string_compose_sp_kernel): usage tools/pmc_batch.sh <tag>
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
pass() { local name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $name -- python $R/tools/batch_repeat.py 1000000 6 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
pass l2 TCC_HIT_sum TCC_MISS_sum
pass fetch FETCH_SIZE
cd $R
python - <<PY
import glob, sqlite3, os
rows = {}
for db in sorted(glob.glob(os.path.join("$OUT", "*_results.db"))):
    c = sqlite3.connect(db)
    for name, cname, n, tot in c.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events where name like '%string_compose_sp_kernel%' group by name, counter_name"):
        rows[cname] = (n, tot / n)
print("string_compose_sp_kernel, per launch (64 waves x 201 levels):")
for k, (n, v) in sorted(rows.items()):
    print(f"  {k:24s} launches {n:3d}  per launch {v:14.1f}  per wave-level {v/64/201:10.2f}")
PY
