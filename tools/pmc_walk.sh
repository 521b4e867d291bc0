#!/bin/bash
# SQ counters of k_walk_runs / the scan kernel for several builds (analysis only; TCC/TCP passes are slow, left out)
#   LIBS="a.so b.so" tools/pmc_walk.sh [bench args]
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
for lib in ${LIBS:-libacmi.so}; do
for set in "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
  rm -rf /tmp/pmc_x
  ACMI_LIBRARY=$REPO/php_aho_corasick_amd/$lib timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_x -o p -- python $REPO/bench.py --no-cpu --steps 2 --warmup 1 "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc_x -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$lib" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    for t in ("k_walk_runs", "k_scan_anchor", "k_scan_lds"):
        if t in k:
            acc[t][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[t].add(r["Dispatch_Id"])
for t in acc:
    print(sys.argv[2], t, "  ".join("%s=%.3g" % (c, v / max(1, len(cnt[t]))) for c, v in sorted(acc[t].items())))
PY
done
done
