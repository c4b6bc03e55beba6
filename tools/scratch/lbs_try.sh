#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4w
timeout 600 python -m pytest tests/test_gpu_deform.py tests/test_gpu_api.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -n 3
for i in 1 2; do
rm -rf gpurun_out/r4w/hl
timeout 300 rocprofv3 --kernel-trace -f rocpd -d gpurun_out/r4w/hl -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --metric-only > gpurun_out/r4w/hl.log 2>&1
python tools/timeline.py $(find gpurun_out/r4w/hl -name "*_results.db" | head -1) 2>&1 | grep -E "period|lbs"
done
rm -rf gpurun_out/r4w/c5
timeout 200 rocprofv3 --kernel-trace -f rocpd -d gpurun_out/r4w/c5 -o t -- python tools/config_timeline.py C5 > gpurun_out/r4w/c5.log 2>&1
python tools/timeline.py $(find gpurun_out/r4w/c5 -name "*_results.db" | head -1) 2>&1 | grep -E "period|lbs"
find gpurun_out/r4w -name "*.db" -delete
