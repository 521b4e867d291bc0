#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace) for several builds of libacmi in one gpurun call
#   LIBS="a.so b.so" tools/ab_trace.sh [bench.py args]
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
for lib in $LIBS; do
  rm -rf /tmp/tr_$lib
  ACMI_LIBRARY=$REPO/php_aho_corasick_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$lib -o t -- python $REPO/bench.py --no-cpu "$@" > /dev/null 2>&1
  f=$(find /tmp/tr_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib $*"; grep -E "k_scan_anchor|k_walk_runs|k_scan_lds|k_expand" $f | sed -E 's/^"(void )?acmi::([a-z_]+)[^"]*",([0-9]+),[0-9]+,([0-9.]+),.*/\2 calls \3 avg_ns \4/'
  t=$(find /tmp/tr_$lib -name "*kernel_trace.csv" | head -1)
  python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: print the sequence of kernels with start offsets and durations
idx = [i for i, r in enumerate(rows) if "k_scan_anchor" in r["Kernel_Name"] or "k_scan_lds" in r["Kernel_Name"]]
i0 = idx[-1] - 8
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:idx[-1] + 14]:
    print("   %-28s start %8.1f us  dur %8.1f us" % (r["Kernel_Name"].split("(")[0][-28:], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
done
