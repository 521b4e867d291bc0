"""Condense rocprofv3 output directories (made by tools/prof.sh) into one text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows:
        print(f"{r.get('Name','')[:70]:70s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} "
              f"avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")
print()
# The default bench command also runs the secondary legs, which launch the same kernel templates on smaller batches:
# the per-kernel average above mixes them.  Per (kernel, grid size) the launches of one workload stand alone.
print("== acmi kernels by launch shape (same trace): calls, mean / min / max duration in us ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    groups = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            if "acmi::" not in name:
                continue
            key = (name.replace("void ", "")[:64], int(r.get("Grid_Size_X", 0)), int(r.get("Workgroup_Size_X", 0)))
            groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (name, grid, wg), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        if sum(d) < 200.0:
            continue
        print(f"{name:64s} grid={grid:>10d} wg={wg:>5d} calls={len(d):>4d} mean={sum(d) / len(d):9.1f} min={min(d):9.1f} max={max(d):9.1f}")
print()
print("== PMC counters, summed over dispatches of each kernel (per-dispatch mean in brackets) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")[:60]
                acc[k][r.get("Counter_Name")] += float(r.get("Counter_Value", 0))
                cnt[k].add(r.get("Dispatch_Id"))
        for k, cs in acc.items():
            if not any(t in k for t in ("k_scan_lds", "k_scan_anchor", "k_walk_runs", "k_walk_items", "k_walk_tiles", "k_scan_simple", "k_expand", "k_refill", "k_stream_read", "k_dense_", "k_items", "k_merge", "k_replace")):
                continue
            n = max(1, len(cnt[k]))
            print(f"{k}: dispatches={n}")
            for c, v in sorted(cs.items()):
                print(f"    {c:28s} {v:18.0f}  [{v / n:16.1f}]")
print()
for f in sorted(glob.glob(os.path.join(out, "bench_*.log"))):
    with open(f) as fh:
        for line in fh:
            if line.startswith("{"):
                print(os.path.basename(f), line.strip()[:400])
