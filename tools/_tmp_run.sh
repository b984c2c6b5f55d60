export TMPDIR=/tmp
mkdir -p gpurun_out/tail
for o in s2-first s1-first s2-first s1-first; do
echo "order $o"
timeout 300 python bench.py --no-extras --no-cpu-baseline --order $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','step_host_phases_us')}, d['ms_per_step_stats']['std'], d['ms_per_step_stats']['p99'])"
done | tee gpurun_out/tail/bench_order.txt
