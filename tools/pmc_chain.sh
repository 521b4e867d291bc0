#!/bin/bash
# Kernel trace + two SQ counter passes of one chain-walk cell (tools/bench_density.py, kernel 6, 1 GiB); per-dispatch means.
#   tools/pmc_chain.sh <tag> <set> <per 8 KiB> [option=value ...]         (ACMI_LIBRARY picks the build)
TAG=$1; SET=$2; PER=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
CMD="python $REPO/tools/bench_density.py $SET $PER 8192 mib=1024 calls=2 kernel=6 $*"
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o pmc -- $CMD > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o pmc -- $CMD > $OUT/pmc2.log 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/trace_kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:6]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
for p in ("pmc1", "pmc2"):
    for f in glob.glob(out + f"/{p}/**/pmc_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            if "chain" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            print(k, "dispatches", len(n[k]), {c: round(x / len(n[k]) / 1e6, 2) for c, x in v.items()}, "(M per dispatch)")
PY
