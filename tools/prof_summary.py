#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs that tools/prof.sh wrote: kernel stats + per-kernel counter sums."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        lines.append("%-60s calls=%s avg_ns=%s total_ns=%s pct=%s" % (row["Name"][:60], row["Calls"], row["AverageNs"], row["TotalDurationNs"], row["Percentage"]))
for p in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        n = defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        lines.append("== counters, summed over dispatches (%s)" % os.path.relpath(f, out))
        for k, d in acc.items():
            if "render" not in k:
                continue
            for c, v in sorted(d.items()):
                lines.append("%-40s %-28s %.6g" % (k, c, v))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
