export TMPDIR=/tmp
mkdir -p gpurun_out/tail
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -x -q -m gpu -k "batch or config3 or fused or string or rccl" 2>&1 | tail -5 | tee gpurun_out/tail/tests4.txt
timeout 300 python tools/step_breakdown.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tail/step_breakdown4.txt
WFST_STRING_UNPACKED=1 timeout 300 python tools/step_breakdown.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/tail/step_breakdown4.txt
timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','ms_shortest_path_T','ms_compose_shortest_path_batch','step_host_phases_us')}, d['ms_per_step_stats'])" | tee gpurun_out/tail/bench4.txt
