#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -n 3 > gpurun_out/final/pytest_gpu.txt
cat gpurun_out/final/pytest_gpu.txt
bash tools/profile_round.sh final > gpurun_out/final/profile_round.log 2>&1
bash tools/profile_extras.sh final > gpurun_out/final/profile_extras.log 2>&1
timeout 600 python bench.py > gpurun_out/final/bench2.json 2> gpurun_out/final/bench2.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/final/bench2.json').read().strip().split('\n')[-1])
print(b['value'], b['ms_per_step'], b['train_step']['ms_per_step'], b['dense_gradient_scene']['ms_per_step'], b['cycling_cameras']['ms_per_step'])
PY
head -3 gpurun_out/final/graph_timeline.txt; cat gpurun_out/final/configs.txt | grep "^C"
