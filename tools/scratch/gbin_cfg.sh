#!/bin/bash
# every configuration with the direct and with the grouped tile sort
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || exit 1
for m in 0 1; do echo "== RIGGS_BIN_GROUPED=$m"; RIGGS_BIN_GROUPED=$m timeout 400 python tools/configs_sweep.py 2>&1 | grep -E "^C[1-5]"; done
timeout 300 python -m pytest tests/test_gpu_raster.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -n 2
RIGGS_BIN_GROUPED=1 timeout 300 python -m pytest tests/test_gpu_raster.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -n 2
