#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs that tools/prof.sh wrote: kernel stats + per-kernel counter sums (summary.txt), and the
render kernel's numbers PER LAUNCH as pmc.json (what bench.py reads from profiles/ for its executed-work figures)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
js = {"kernel": None, "launches": None, "avg_ns": None, "counters_per_launch": {}}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        lines.append("%-60s calls=%s avg_ns=%s total_ns=%s pct=%s" % (row["Name"][:60], row["Calls"], row["AverageNs"], row["TotalDurationNs"], row["Percentage"]))
        if "render_kernel" in row["Name"] and js["kernel"] is None:
            js.update(kernel=row["Name"].split("(")[0], launches=int(row["Calls"]), avg_ns=float(row["AverageNs"]))
for p in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
        lines.append("== counters, summed over dispatches (%s)" % os.path.relpath(f, out))
        for k, d in acc.items():
            if "render" not in k and "tail_sum" not in k:
                continue
            for c, v in sorted(d.items()):
                lines.append("%-40s %-28s %.6g  (%d dispatches)" % (k, c, v, len(disp[k])))
                if "render" in k:
                    js["counters_per_launch"][c] = js["counters_per_launch"].get(c, 0.0) + v / max(len(disp[k]), 1)
                elif c in ("FETCH_SIZE", "WRITE_SIZE"):   # the frame's second kernel (sums the tail's colours): its HBM traffic belongs to the frame
                    js["counters_per_launch"][c] = js["counters_per_launch"].get(c, 0.0) + v / max(len(disp[k]), 1)
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(js, open(os.path.join(out, "pmc.json"), "w"), indent=1)
print("\n".join(lines))
