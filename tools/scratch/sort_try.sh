#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4w
timeout 400 python tools/configs_sweep.py 2>&1 | grep -E "^C[5]"
rm -rf gpurun_out/r4w/c5
timeout 200 rocprofv3 --kernel-trace -f rocpd -d gpurun_out/r4w/c5 -o t -- python tools/config_timeline.py C5 > gpurun_out/r4w/c5.log 2>&1
python tools/timeline.py $(find gpurun_out/r4w/c5 -name "*_results.db" | head -1) > gpurun_out/r4w/c5_timeline.txt 2>&1
grep -E "period|rs_" gpurun_out/r4w/c5_timeline.txt
find gpurun_out/r4w -name "*.db" -delete
