#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4w
timeout 600 python -m pytest tests/test_gpu_deform.py tests/test_gpu_api.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -n 3
timeout 400 python tools/configs_sweep.py 2>&1 | grep -E "^C[45]"
rm -rf gpurun_out/r4w/c5
timeout 200 rocprofv3 --kernel-trace -f rocpd -d gpurun_out/r4w/c5 -o t -- python tools/config_timeline.py C5 > gpurun_out/r4w/c5.log 2>&1
python tools/timeline.py $(find gpurun_out/r4w/c5 -name "*_results.db" | head -1) 2>&1 | grep -E "period|lbs"
find gpurun_out/r4w -name "*.db" -delete
