#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats + separate PMC passes (never combined with
# other trace domains).  Usage (through gpurun): bash tools/prof.sh <tag> [bench args...]
# Outputs land in gpurun_out/<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-prof}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 3 --warmup 1 --no-cpu --no-dense --no-legs $*"
# the judged command is the DEFAULT bench invocation (CPU baseline leg included)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $REPO/bench.py $* > "$OUT/bench_trace.log" 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$n" -o pmc -- python $REPO/bench.py $ARGS > "$OUT/bench_pmc_$n.log" 2>&1
done
cd "$REPO"
python tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
