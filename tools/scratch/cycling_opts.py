"""bench.cycling_cameras_timing under forward work-list options: is the previous frame's walk history worth anything when every replay has another camera?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from riggs_amd import _lib as L
for tiles in (256, 0, 256, 0):
    L.set_option("fwd_wide_tiles", tiles)
    r = bench.cycling_cameras_timing("cuda:0", steps=128)
    print("fwd_wide_tiles=%d: %.4f ms" % (tiles, r["ms_per_step"]), flush=True)
