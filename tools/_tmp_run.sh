export TMPDIR=/tmp
mkdir -p gpurun_out/soak
timeout 300 python tools/soak.py 100 900000 2>&1 | tail -2 | tee gpurun_out/soak/soak.txt
timeout 300 python tools/soak_sssp.py 60 500000 2>&1 | tail -2 | tee gpurun_out/soak/soak_sssp.txt
timeout 300 python tools/soak_ops.py 40 300000 2>&1 | tail -2 | tee gpurun_out/soak/soak_ops.txt
