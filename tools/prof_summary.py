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
# PER LAUNCH means per rt_render call (per frame): with rt_set_tile_order a frame is two launches of render_kernel (the measuring pass
# of one sample, then the rest; they may be different instances of the template) and one of tail_sum_kernel, so the frames are counted
# by tail_sum_kernel's launches where there are any, and render_kernel's times and counters are summed over its instances.
js = {"kernel": None, "launches": None, "avg_ns": None, "counters_per_launch": {}}
render_total_ns, render_calls, tail_calls, biggest = 0.0, 0, 0, 0.0
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        lines.append("%-60s calls=%s avg_ns=%s total_ns=%s pct=%s" % (row["Name"][:60], row["Calls"], row["AverageNs"], row["TotalDurationNs"], row["Percentage"]))
        if "render_kernel" in row["Name"]:
            if float(row["TotalDurationNs"]) > biggest:
                biggest = float(row["TotalDurationNs"]); js["kernel"] = row["Name"].split("(")[0]
            render_total_ns += float(row["TotalDurationNs"]); render_calls += int(row["Calls"])
        if "tail_sum_kernel" in row["Name"]:
            tail_calls += int(row["Calls"])
if render_calls:
    frames = tail_calls if tail_calls else render_calls
    js.update(launches=frames, avg_ns=render_total_ns / frames, render_kernel_launches=render_calls)
    lines.append("== render_kernel per frame: %d frames, %d launches, %.0f ns per frame" % (frames, render_calls, render_total_ns / frames))
for p in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for f in glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
        lines.append("== counters, summed over dispatches (%s)" % os.path.relpath(f, out))
        n_tail = sum(len(v) for k, v in disp.items() if "tail_sum" in k)
        n_render = sum(len(v) for k, v in disp.items() if "render" in k)
        frames = n_tail if n_tail else max(n_render, 1)
        for k, d in acc.items():
            if "render" not in k and "tail_sum" not in k:
                continue
            for c, v in sorted(d.items()):
                lines.append("%-40s %-28s %.6g  (%d dispatches, %d frames)" % (k, c, v, len(disp[k]), frames))
                if "render" in k:
                    js["counters_per_launch"][c] = js["counters_per_launch"].get(c, 0.0) + v / frames
                elif c in ("FETCH_SIZE", "WRITE_SIZE"):   # the frame's second kernel (sums the tail's colours): its HBM traffic belongs to the frame
                    js["counters_per_launch"][c] = js["counters_per_launch"].get(c, 0.0) + v / frames
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(js, open(os.path.join(out, "pmc.json"), "w"), indent=1)
print("\n".join(lines))
