#!/bin/bash
# A/B of bench.py step schedules (runs on the GPU box): prints ms_per_step etc. for each variant
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'], 'std', d['ms_per_step_stats']['std'], 'sp(T) alone', d['ms_shortest_path_T'], 'batch alone', d['ms_compose_shortest_path_batch'])"; }
for p in "0 -1" "-1 0" "0 0" "-1 -1"; do set -- $p; WFST_BENCH_PRIO1=$1 WFST_BENCH_PRIO2=$2 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | show "mailbox prio sssp=$1 batch=$2"; done
