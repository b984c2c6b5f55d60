# GPU box: parity of the wide composition driver + its timings (tools/lookahead_timing.py) + a kernel trace
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/wide2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide or lookahead or large_pair or config5" > gpurun_out/wide2/tests.log 2>&1
tail -3 gpurun_out/wide2/tests.log
timeout 600 python tools/lookahead_timing.py 300,20,3,8,8 2000,50,3,12,12 10000,100,3,16,16 40000,100,3,16,16 2>&1 | grep "^ *[0-9n]" | tee gpurun_out/wide2/timing.txt
WFST_WIDE_NO_FORESIGHT=1 timeout 300 python tools/lookahead_timing.py 10000,100,3,16,16 40000,100,3,16,16 2>&1 | grep "^ *[0-9]" | tee gpurun_out/wide2/noforesight.txt
WFST_WIDE_TRACE=1 timeout 300 python tools/lookahead_timing.py 10000,100,3,16,16 2>&1 | grep "^wide" | tail -24 > gpurun_out/wide2/trace_1m.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wide2 -o wide -- python $R/tools/lookahead_timing.py 10000,100,3,16,16 > $R/gpurun_out/wide2/wide_trace.log 2>&1
