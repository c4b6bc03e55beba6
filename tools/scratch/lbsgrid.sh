#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4w
for cap in 1280 640 320; do
rm -rf gpurun_out/r4w/hl
RIGGS_LBS_MAX_GRID=$cap timeout 300 rocprofv3 --kernel-trace -f rocpd -d gpurun_out/r4w/hl -o t -- python tools/config_timeline.py C3 > gpurun_out/r4w/hl.log 2>&1
echo "== cap $cap (C3)"; python tools/timeline.py $(find gpurun_out/r4w/hl -name "*_results.db" | head -1) 2>&1 | grep -E "period|lbs_forward"
rm -rf gpurun_out/r4w/hl
RIGGS_LBS_MAX_GRID=$cap timeout 300 rocprofv3 --kernel-trace -f rocpd -d gpurun_out/r4w/hl -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --metric-only > gpurun_out/r4w/hl.log 2>&1
echo "== cap $cap (headline)"; python tools/timeline.py $(find gpurun_out/r4w/hl -name "*_results.db" | head -1) 2>&1 | grep -E "period|lbs_forward"
done
find gpurun_out/r4w -name "*.db" -delete
