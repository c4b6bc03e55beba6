#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -n 3
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -n 1 | cut -c1-400
timeout 400 python tools/configs_sweep.py 2>&1 | grep -E "^C[1-5]"
