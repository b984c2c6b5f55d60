#!/bin/bash
# GPU box: the result exchange through the C-ABI with a 1-rank RCCL group: test output (begin / end host cost) and the bench
# step with and without it
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-dist}
mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_distributed.py -x -q -m gpu -s > $OUT/rccl_test.txt 2>&1; grep -a "C-ABI gather\|passed\|failed" $OUT/rccl_test.txt
timeout -k 5 200 python bench.py --no-extras --no-cpu-baseline --steps 500 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 WFST_BENCH_FORCE_DIST=1 timeout -k 5 200 python bench.py --no-extras --no-cpu-baseline --steps 500 > $OUT/bench_dist.json 2> $OUT/bench_dist.err
python - <<PY
import json
for n in ("plain", "dist"):
    try:
        d = json.loads(open("$OUT/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "ms_per_step", d["ms_per_step"], d["ms_per_step_stats"], "rccl_used", d["rccl_used"])
    except Exception as e:
        print(n, "failed", e, open("$OUT/bench_%s.err" % n).read()[-800:])
PY
