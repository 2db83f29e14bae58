"""tools/trace_gaps.py TRACE.csv — per-kernel average duration and the idle gap before each kernel
(start - previous end) from a rocprofv3 --kernel-trace CSV.  Prints a small JSON summary."""
import csv, json, re, sys, collections

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"]) or re.search(r"\w+", r["Kernel_Name"])).group(0)))
rows.sort()
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
prev_end = None
for s, e, k in rows:
    dur[k].append(e - s)
    if prev_end is not None:
        gap[k].append(s - prev_end)
    prev_end = e
out = {}
for k, v in dur.items():
    if len(v) < 50:
        continue
    g = sorted(gap[k])
    out[k] = {"calls": len(v), "avg_us": sum(v) / len(v) / 1e3, "median_us": sorted(v)[len(v) // 2] / 1e3,
              "gap_before_median_us": g[len(g) // 2] / 1e3 if g else None,
              "gap_before_avg_us": sum(g) / len(g) / 1e3 if g else None}
print(json.dumps(out, indent=1))
