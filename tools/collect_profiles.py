"""Copy what is to be judged from a gpurun_out/<tag>/ profile directory (tools/prof.sh, tools/prof_cfg3.sh) into
profiles/<round>/: the kernel-stats CSV, the text summary, and per PMC pass the counter rows of this library's kernels.
    python tools/collect_profiles.py gpurun_out/r02_final profiles/r02 [prefix_]"""
import csv
import glob
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
prefix = sys.argv[3] if len(sys.argv) > 3 else ""
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, prefix + "kernel_stats.csv"))
if os.path.exists(os.path.join(src, "summary.txt")):
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, prefix + "rocprofv3_summary.txt"))
KEEP = ("k_scan_lds", "k_scan_anchor", "k_walk_runs", "k_walk_items", "k_stream_read", "k_refill_wave", "k_expand", "k_items", "k_rec_scan", "k_dense_decide", "k_dense_emit", "k_merge", "k_replace", "k_scan_u64", "k_walk_tiles", "k_match_small", "k_gather", "k_walk_chain", "k_chain_emit", "k_runs_wave", "k_expand_dense", "k_walk_win")
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows = list(csv.reader(fh))
        head, body = rows[0], rows[1:]
        k = head.index("Kernel_Name")
        body = [r for r in body if any(t in r[k] for t in KEEP)]
        # at most 6 dispatches per kernel keep the files small
        seen, out = {}, []
        di = head.index("Dispatch_Id")
        for r in body:
            ids = seen.setdefault(r[k], [])
            if r[di] not in ids:
                if len(ids) >= 6:
                    continue
                ids.append(r[di])
            out.append(r)
        with open(os.path.join(dst, prefix + os.path.basename(d) + ".csv"), "w", newline="") as fh:
            w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(head)
            w.writerows(out)
print(sorted(os.listdir(dst)))
