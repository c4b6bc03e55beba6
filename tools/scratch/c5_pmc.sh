#!/bin/bash
# PMC passes over the C5 frame (graph replays): what the grouped tile sort's kernels wait for
export TMPDIR=/tmp
O=$PWD/gpurun_out/c5pmc
mkdir -p $O
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" WRITE_SIZE FETCH_SIZE; do
  D=$O/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $D -o c -- python tools/config_timeline.py C5 > $D.log 2>&1
  tail -n 2 $D.log | cut -c1-200
done
python tools/kernel_counters.py $O/kernel_counters.json c5 $O/pmc_* > $O/kernel_counters.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/c5pmc/kernel_counters.json'))
for k in ("gbin_count_kernel","gbin_scatter_kernel","gbin_tcount_kernel","gbin_tscatter_kernel","rs_scatter_kernel","lbs_forward_kernel"):
    if k in d: print(k, json.dumps(d[k]))
PY
find $O -name "*.db" -delete
