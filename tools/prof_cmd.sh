#!/bin/bash
# usage: tools/prof_cmd.sh <name> <command ...>   kernel statistics (rocprofv3 --kernel-trace --stats) of any command, top 14 kernels
name=$1; shift
repo=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$repo/gpurun_out/$name
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- "$@" > $out/trace.log 2>&1
grep -v "^W\|^E\|amdgpu.ids" $out/trace.log
python - <<PY
import csv,glob
f=glob.glob("$out/**/trace_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
