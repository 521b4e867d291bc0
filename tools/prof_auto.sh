#!/bin/bash
# kernel trace of tools/exp_auto.py <kind> <level> [kernel]  ->  gpurun_out/<tag>/
TAG=$1; shift
R=$(pwd); mkdir -p $R/gpurun_out/$TAG; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o t -- python $R/tools/exp_auto.py "$@" > $R/gpurun_out/$TAG/log.txt 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/$TAG/t_kernel_stats.csv")):
    if "acmi" in r["Name"]: print("  ", r["Name"][:48], r["Calls"], round(float(r["AverageNs"])/1000,1), "us")
PY
