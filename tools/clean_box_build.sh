#!/bin/bash
# Proof that the tree builds from SOURCE on a GPU box: every built artefact removed first, then __graft_entry__.build() (every
# translation unit, verbose), smoke(), and the whole GPU suite on the freshly built libraries.
# usage (repo root, on the GPU box): bash tools/clean_box_build.sh > gpurun_out/clean_box_build.txt 2>&1
echo "== built artefacts removed on the box before anything runs =="
rm -rf riggs_amd/lib oracle/_build oracle/_ref
ls riggs_amd/lib oracle/_build 2>&1
echo "== toolchain =="
/opt/rocm/bin/hipcc --version | head -2
echo "host cores: $(nproc)"
echo "== __graft_entry__.build(): every translation unit from source (riggs_amd.build verbose) =="
python - <<'PY'
import time
t0 = time.perf_counter()
from riggs_amd import build as B
so = B.build(force=True, verbose=True)
import os
print("libriggs_hip.so built in %.1f s: %s (%d bytes)" % (time.perf_counter() - t0, so, os.path.getsize(so)))
t1 = time.perf_counter()
import __graft_entry__ as g
g.build()
print("__graft_entry__.build() (PyTorch extension front-end, oracle, symbol check) %.1f s more" % (time.perf_counter() - t1))
PY
ls -l oracle/_build/*.so riggs_amd/lib/*.so
echo "objects: $(ls riggs_amd/lib/obj/*.o | wc -l)"
echo "exported riggs_* functions: $(nm -D riggs_amd/lib/libriggs_hip.so | grep -c ' T riggs_')"
echo "== smoke =="
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== gpu tests on the freshly built libraries =="
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
