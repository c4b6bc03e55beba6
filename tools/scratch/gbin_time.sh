#!/bin/bash
# phase timers of the grouped binning kernels at C5 (device printf from a few workgroups)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -ffp-contract=fast"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4w
/opt/rocm/bin/hipcc -c riggs_amd/csrc/binning.hip -o riggs_amd/lib/obj/binning.o $FLAGS -DBIN_GROUPED_MIN_T=4096 -DGB_TIME || exit 1
/opt/rocm/bin/hipcc -shared -o riggs_amd/lib/libriggs_hip.so --offload-arch=gfx950 riggs_amd/lib/obj/*.o
timeout 200 python tools/config_timeline.py C5 > gpurun_out/r4w/c5t.log 2>&1
grep GBS gpurun_out/r4w/c5t.log | tail -n 10
grep GTS gpurun_out/r4w/c5t.log | tail -n 14
