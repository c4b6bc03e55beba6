#!/bin/bash
# A/B runs of render.hip built with -DFW_EXP=<n> (on the GPU box: the library in the snapshot is rebuilt per variant).
# usage: bash tools/variants.sh <out file> <n> [<n> ...]
OUT=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -ffp-contract=fast"
for v in "$@"; do
  /opt/rocm/bin/hipcc -c riggs_amd/csrc/render.hip -o riggs_amd/lib/obj/render.o $FLAGS -DFW_EXP=$v 2>> $OUT.err || { echo "variant $v: build failed" >> $OUT; continue; }
  /opt/rocm/bin/hipcc -shared -o riggs_amd/lib/libriggs_hip.so --offload-arch=gfx950 riggs_amd/lib/obj/*.o
  python tools/fwd_time.py "variant $v" >> $OUT 2>> $OUT.err
done
cat $OUT
