"""Timeline of the last full pipeline step from a rocprofv3 kernel trace csv (analysis only)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_scan_anchor" in r["Kernel_Name"] or "k_scan_lds" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]   # two consecutive steps
def first_of_step(i):
    j = i
    while j > 0 and "fillBuffer" not in rows[j]["Kernel_Name"]:
        j -= 1
    while j > 0 and "fillBuffer" in rows[j - 1]["Kernel_Name"]:
        j -= 1
    return j
s0, s1 = first_of_step(a), first_of_step(b)
t0 = int(rows[s0]["Start_Timestamp"])
busy = 0
prev_end = None
for r in rows[s0:s1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    busy += en - st
    print("  %-30s start %8.1f  dur %7.1f  gap before %6.1f" % (r["Kernel_Name"].split("(")[0][-30:], (st - t0) / 1e3, (en - st) / 1e3, gap))
    prev_end = en
period = (int(rows[s1]["Start_Timestamp"]) - t0) / 1e3
print("step period %.1f us, kernels busy %.1f us, idle %.1f us" % (period, busy / 1e3, period - busy / 1e3))
