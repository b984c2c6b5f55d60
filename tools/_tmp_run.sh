export TMPDIR=/tmp
mkdir -p gpurun_out/tail
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail_in_one_launch or repeated_queries or async_shortest or shortest_path" 2>&1 | tail -5 | tee gpurun_out/tail/tests.txt
for v in 0 1; do
  if [ $v = 1 ]; then export WFST_SSSP_SPLIT_TAIL=1; fi
  timeout 300 python tools/sp_repeat.py 1000000 200 2>&1 | tail -3
done | tee gpurun_out/tail/sp_repeat.txt
unset WFST_SSSP_SPLIT_TAIL
timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','ms_shortest_path_T','ms_compose_shortest_path_batch')})" | tee gpurun_out/tail/bench.txt
