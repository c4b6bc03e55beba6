#!/bin/bash
# PMC passes over one configuration's frame (graph replays), one pass per counter group, merged per kernel by
# tools/kernel_counters.py: bytes fetched / written against the algorithmic ones, write-request sizes, wait cycles.
# This is what showed the tile sorts gathering 240 MB for 32 MB of rectangles, the depth sort writing 118 MB for 16 MB
# and the skinning backward waiting 71 % of its cycles at one workgroup per CU (profiles/round3_C5_kernel_counters_*.json).
# usage (from the repo root, on the GPU box): bash tools/config_counters.sh [C5]      -> gpurun_out/pmc_<config>/kernel_counters.json
CFG=${1:-C5}
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc_$CFG
mkdir -p $O
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" WRITE_SIZE FETCH_SIZE; do
  D=$O/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $D -o c -- python tools/config_timeline.py $CFG > $D.log 2>&1
  tail -n 2 $D.log | cut -c1-200
done
python tools/kernel_counters.py $O/kernel_counters.json $CFG $O/pmc_* > $O/kernel_counters.log 2>&1
O_DIR=$O python - <<'PY'
import json
import os
d=json.load(open(os.path.join(os.environ.get('O_DIR','gpurun_out/pmc_C5'),'kernel_counters.json')))
print("%-30s %9s %8s %8s %9s %9s" % ("kernel", "rdMB(x2)", "wrMB", "VALU M", "wr<64B M", "wr64B M"))
for k, v in d.items():
    if k.startswith('_'): continue
    print("%-30s %9.0f %8.0f %8.1f %9.2f %9.2f" % (k, v.get('read_bytes_x2_corrected', 0) / 1e6, v.get('write_bytes', 0) / 1e6, v.get('SQ_INSTS_VALU', 0) / 1e6,
          (v.get('TCC_EA0_WRREQ_sum', 0) - v.get('TCC_EA0_WRREQ_64B_sum', 0)) / 1e6, v.get('TCC_EA0_WRREQ_64B_sum', 0) / 1e6))
PY
find $O -name "*.db" -delete
