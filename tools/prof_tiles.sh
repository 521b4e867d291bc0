#!/bin/bash
# Counters of k_walk_tiles (kernel 6) on one tools/bench_tiles.py case, kernel 6 forced (ONLY6=1 for short8, K=6 for the
# headlineNN cases).  Usage (through gpurun): bash tools/prof_tiles.sh <tag> <case>   -> gpurun_out/<tag>/summary.txt
set -u
TAG=${1:-tiles}; CASE=${2:-short8}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp ONLY6=1 K=6
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $REPO/tools/bench_tiles.py $CASE > "$OUT/bench_trace.log" 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "WRITE_SIZE SQ_WAVES"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$n" -o pmc -- python $REPO/tools/bench_tiles.py $CASE > "$OUT/bench_pmc_$n.log" 2>&1
done
cd "$REPO"
python tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
