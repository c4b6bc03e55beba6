#!/bin/bash
# same-box A/B: the tree under .ab_base (one wave per block in preprocess_bwd_lean) against this tree (two waves)
for i in 1 2; do
  for d in .ab_base .; do
    (cd $d && python - <<EOF
import bench, json, torch
r = bench.dense_scene_timing("cuda:0")
c = bench.cycling_cameras_timing("cuda:0", steps=64)
print("$d dense", r["ms_per_step"], r.get("kernels_us", {}).get("preprocess_bwd"), "cycling", c if not isinstance(c, dict) else c.get("ms_per_step"), flush=True)
EOF
    ) 2>/dev/null | tail -1
  done
done
