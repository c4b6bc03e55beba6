"""Graph replays of the headline workload with a new camera every replay (bench.cycling_cameras_timing) for a kernel trace:
  rocprofv3 --kernel-trace --stats -f csv -d out -o t -- python tools/cycling_timeline.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(bench.cycling_cameras_timing("cuda:0", steps=64))
