#!/bin/bash
# variants of the second level at C5 (grouped binning forced on): timeline per variant
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -ffp-contract=fast"
export TMPDIR=/tmp
mkdir -p gpurun_out/r4w
for v in "-DGTS_THREADS=1024 -DGTS_WAVES=8" "-DGTS_THREADS=1024 -DGTS_WAVES=4" "-DGTS_THREADS=512 -DGTS_WAVES=4" "-DGTS_THREADS=512 -DGTS_WAVES=6"; do
/opt/rocm/bin/hipcc -c riggs_amd/csrc/binning.hip -o riggs_amd/lib/obj/binning.o $FLAGS -DBIN_GROUPED_MIN_T=4096 $v || exit 1
/opt/rocm/bin/hipcc -shared -o riggs_amd/lib/libriggs_hip.so --offload-arch=gfx950 riggs_amd/lib/obj/*.o
rm -rf gpurun_out/r4w/c5
timeout 200 rocprofv3 --kernel-trace -f rocpd -d gpurun_out/r4w/c5 -o t -- python tools/config_timeline.py C5 > gpurun_out/r4w/c5.log 2>&1
echo "== $v"
python tools/timeline.py $(find gpurun_out/r4w/c5 -name "*_results.db" | head -1) 2>&1 | grep -E "period|gbin"
done
timeout 300 python -m pytest tests/test_gpu_raster.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -n 2
find gpurun_out/r4w -name "*.db" -delete
