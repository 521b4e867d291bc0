#!/bin/bash
# kernel-trace stats only: bash tools/trace.sh <tag> [bench args]
TAG=${1:-trace}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu "$@" > "$OUT/bench.log" 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv,glob,sys,os
out=sys.argv[1]
for f in glob.glob(os.path.join(out,"trace","**","*kernel_stats.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} pct={r['Percentage']}")
print(open(os.path.join(out,"bench.log")).read()[-900:])
PY
