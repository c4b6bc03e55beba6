"""Opaque-skin scene: preprocess_bwd as the lean form (default with sparse rows) against the 256-thread form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from riggs_amd import _lib as L
for rep in range(2):
    for v in (-1, 0):
        L.set_option("preprocess_bwd_lean", v)
        r = bench.dense_scene_timing("cuda:0")
        print("preprocess_bwd_lean", v, "dense", r["ms_per_step"], flush=True)
L.set_option("preprocess_bwd_lean", -1)
