#!/bin/bash
# same-box A/B of the heads-on captured iteration over several trees: heads_ab3.sh tree...
for i in 1 2; do
  for d in "$@"; do
    (cd $d && python - <<EOF
import bench, json
r = bench.train_step_heads_timing("cuda:0")
print("$d", r["ms_per_step"], r["heads_off_ms_per_step"], r["heads_ms"], flush=True)
EOF
    ) 2>/dev/null | tail -1
  done
done
