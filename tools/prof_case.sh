#!/bin/bash
# usage: tools/prof_case.sh <name> <bench_tiles args...>   kernel statistics of one bench_tiles case (env passes through)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python $GRAFT_REPO_ROOT/tools/bench_tiles.py "$@" > $out/trace.log 2>&1
grep -v "^W\|^E" $out/trace.log
python - <<PY
import csv,glob
f=glob.glob("$out/**/trace_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
