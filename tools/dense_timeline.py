"""Graph replays of the DENSE-GRADIENT scene (bench.dense_scene_timing) for a kernel trace:
  rocprofv3 --kernel-trace -f rocpd -d out -o t -- python tools/dense_timeline.py ; python tools/timeline.py out/.../t_results.db"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(bench.dense_scene_timing("cuda:0", steps=30))
