#!/bin/bash
# Profile the dense-match drain (BASELINE configs[4], tools/bench_dense.py): kernel-trace stats, then PMC passes
# in runs of their own (a lighter invocation: counter collection serialises dispatches; every rocprofv3 call is under
# `timeout` -- a counter pass that aborted once sat on the box until gpurun's limit).  Usage (through gpurun): bash tools/prof_dense.sh <tag>   -> gpurun_out/<tag>/
set -u
TAG=${1:-dense}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $REPO/tools/bench_dense.py 4096 > "$OUT/bench_trace.log" 2>&1
for set in "WRITE_SIZE GRBM_GUI_ACTIVE" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 180 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$n" -o pmc -- python $REPO/tools/bench_dense.py 1024 8192 light > "$OUT/bench_pmc_$n.log" 2>&1
done
cd "$REPO"
python tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -v amdgpu.ids "$OUT/bench_trace.log" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
