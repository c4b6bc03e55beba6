#!/bin/bash
# Texture-addresser / L1 counters of the compositing kernels (is the forward bound by the line requests of its gathers?)
# usage (GPU box, repo root): bash tools/fwd_memory_counters.sh
export TMPDIR=/tmp
O=$PWD/gpurun_out/fwdmem
mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --metric-only --no-graph"
for C in "TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  D=$O/$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $C -f csv -d $D -o c -- $B > $D.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/c_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("render_fwd", "render_bwd", "preprocess_fwd"):
            if key in k:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in sorted(v.items())}, "(millions per launch)")
PY
