#!/bin/bash
# Counters of the keyword set (bench.py's keyword_set leg: 2,048 patterns of 4..12 letters and spaces over 1 GiB) through the
# dense anchor form: k_scan_anchor<1,..,DENSE,HOT>, k_dense_decide, k_dense_emit.
# Usage (through gpurun): bash tools/prof_words3.sh <tag>   -> gpurun_out/<tag>/summary.txt
set -u
TAG=${1:-words3}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $REPO/tools/exp_dense.py text=kw > "$OUT/bench_trace.log" 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 180 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$n" -o pmc -- python $REPO/tools/exp_dense.py text=kw > "$OUT/log_$n.log" 2>&1
done
cd "$REPO"
python tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
