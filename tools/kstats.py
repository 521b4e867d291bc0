"""Print the top rows of a rocprofv3 --kernel-trace --stats directory: python tools/kstats.py <dir> [n]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
files = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))
for f in files:
    print(f)
    for r in list(csv.DictReader(open(f)))[:n]:
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:10.1f} total_ms {float(r['TotalDurationNs']) / 1e6:9.2f} {r['Percentage']}%")
