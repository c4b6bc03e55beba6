#!/bin/bash
# One profiling session on the GPU box: kernel-trace statistics of `bench.py`, a per-node timeline of one graph
# replay, and the PMC passes (separate runs, --kernel-trace only) that tools/kernel_counters.py merges.
# usage (from the repo root, on the GPU box): bash tools/profile_round.sh <label>      -> gpurun_out/<label>/...
set -u
L=${1:-prof}
O=$PWD/gpurun_out/$L
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --metric-only"
rocprofv3 --kernel-trace --stats -f csv rocpd -d $O/trace -o t -- $B > $O/trace.log 2>&1
DB=$(find $O/trace -name "*_results.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.csv > $O/kernel_stats.log 2>&1 || true
python tools/timeline.py $DB > $O/graph_timeline.txt 2>&1 || true
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  D=$O/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C -f csv -d $D -o c -- $B --no-graph > $D.log 2>&1
done
python tools/kernel_counters.py $O/kernel_counters.json $L $O/pmc_* > $O/kernel_counters.log 2>&1
find $O -name "*.db" -size +20M -delete  # (the merged outputs are what travels back; gpurun_out is capped at 64 MiB)
ls $O
