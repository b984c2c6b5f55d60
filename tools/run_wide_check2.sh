export TMPDIR=/tmp
mkdir -p gpurun_out/wide2
WFST_WIDE_TRACE=1 timeout 300 python tools/lookahead_timing.py 40000,100,3,16,16 2>&1 | grep "^wide\|^ *[0-9]" > gpurun_out/wide2/trace_4m.txt
WFST_WIDE_NO_FORESIGHT=1 timeout 300 python tools/lookahead_timing.py 10000,100,3,16,16 40000,100,3,16,16 2>&1 | grep "^ *[0-9]" > gpurun_out/wide2/noforesight.txt
cat gpurun_out/wide2/noforesight.txt
