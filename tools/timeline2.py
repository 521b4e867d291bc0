"""Print the kernel sequence of the last steps of a rocprofv3 --kernel-trace run with start offsets, durations and gaps (us)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_items" in n]
lo = idx[-3] if len(idx) >= 3 else 0
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = t0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:60]}")
    prev_end = e
