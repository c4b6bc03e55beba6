#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4w; rm -rf gpurun_out/r4w/hl
timeout 300 rocprofv3 --kernel-trace -f rocpd -d gpurun_out/r4w/hl -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --metric-only > gpurun_out/r4w/hl.log 2>&1
python tools/timeline.py $(find gpurun_out/r4w/hl -name "*_results.db" | head -1) > gpurun_out/r4w/hl_timeline.txt 2>&1
cat gpurun_out/r4w/hl_timeline.txt
find gpurun_out/r4w -name "*.db" -delete
