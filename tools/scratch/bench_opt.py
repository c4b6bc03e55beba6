"""Run bench.py with library options set first:  python tools/scratch/bench_opt.py name=value ... -- <bench args>"""
import os
import runpy
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

import riggs_amd._lib as L  # noqa: E402

i = sys.argv.index("--")
for kv in sys.argv[1:i]:
    k, v = kv.split("=")
    L.set_option(k, int(v))
sys.argv = ["bench.py"] + sys.argv[i + 1:]
runpy.run_path("bench.py", run_name="__main__")
