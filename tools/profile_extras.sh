#!/bin/bash
# The rest of a round's evidence, next to tools/profile_round.sh: graph timelines of the dense-gradient scene and of C3 / C5,
# the configuration sweep, the forward's per-wave traces of both scenes, the PoseMLP stage trace.
# usage (from the repo root, on the GPU box): bash tools/profile_extras.sh <label>      -> gpurun_out/<label>/...
set -u
L=${1:-extras}
O=$PWD/gpurun_out/$L
mkdir -p $O
export TMPDIR=/tmp
tl() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace -f rocpd -d $O/$name -o t -- "$@" > $O/$name.log 2>&1
  python tools/timeline.py $(find $O/$name -name "*_results.db" | head -1) > $O/${name}_graph_timeline.txt 2>&1
  find $O/$name -name "*.db" -delete
}
tl dense python tools/dense_timeline.py
tl C3 python tools/config_timeline.py C3
tl C5 python tools/config_timeline.py C5
python tools/configs_sweep.py > $O/configs.txt 2>&1
python tools/fwd_trace.py > $O/fwd_trace_headline.txt 2>&1
python tools/fwd_trace.py dense > $O/fwd_trace_dense.txt 2>&1
python tools/pose_mlp_trace.py > $O/pose_mlp_trace.txt 2>&1
ls $O
