#!/bin/bash
# Counters of the exact whole-item walk (k_walk_items) on the keyword-like set of tools/bench_words.py.
# Usage (through gpurun): bash tools/prof_words.sh <tag>   -> gpurun_out/<tag>/summary.txt
set -u
TAG=${1:-words}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 180 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$n" -o pmc -- python $REPO/tools/bench_words.py light > "$OUT/log_$n.log" 2>&1
done
cd "$REPO"
python tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
