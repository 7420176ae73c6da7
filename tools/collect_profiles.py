#!/usr/bin/env python3
"""Copy the summaries that tools/prof.sh / tools/r2_profile_all.sh left under gpurun_out/prof_<tag>_<workload>/ into profiles/
(<round>_<workload>_summary.txt, _kernel_stats.csv), and rebuild profiles/<round>_pmc.json and profiles/traffic.json, the two files
bench.py reads for `roofline.executed` and `roofline.traffic`.
    python tools/collect_profiles.py <tag> <round>        e.g. r2b r2"""
import glob
import json
import os
import shutil
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
tag, rnd = sys.argv[1], sys.argv[2]
pmc_all, traffic = {}, {}
for w in ("c3", "c2", "box", "c4", "c5"):
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{w}")
    if not os.path.isdir(d):
        continue
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", f"{rnd}_{w}_summary.txt"))
    ks = sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True))
    shutil.copy(ks[-1], os.path.join(ROOT, "profiles", f"{rnd}_{w}_kernel_stats.csv"))
    js = json.load(open(os.path.join(d, "pmc.json")))
    wl = bench.WORKLOADS[w]
    if "SQ_INSTS_VALU" in js["counters_per_launch"]:
        js.update(W=wl["W"], H=wl["H"], spp=wl["spp"], depth=wl["depth"],
                  command=f"tools/prof.sh {tag}_{w}" + ("" if w == "c3" else f" --workload {w}") + " --steps 3 --warmup 1 (rocprofv3 --kernel-trace --stats, then four --pmc passes, each its own run)")
        pmc_all[w] = js
    c = js["counters_per_launch"]
    if "FETCH_SIZE" in c:
        alg = wl["W"] * wl["H"] * 16
        hbm = int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))
        traffic[w] = {"W": wl["W"], "H": wl["H"], "spp": wl["spp"], "n_gpus": 1,
                      "fetch_size_kb_per_launch": round(c["FETCH_SIZE"]), "write_size_kb_per_launch": round(c["WRITE_SIZE"]),
                      "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "hbm_over_algorithmic": round(hbm / alg, 1),
                      "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/prof.sh), render_kernel + tail_sum_kernel, summed over "
                              "the launches / their number; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; source profiles/%s_%s_summary.txt. "
                              "Algorithmic = the float4 accumulator. The rest is the colour buffer of the frame's end (DESIGN.md 4.5b / 4.5d, bounded by "
                              "rt_set_tail_limit, default 1400 MiB), written once by render_kernel in scattered 12-byte stores (partial lines: more write "
                              "traffic than bytes) and read once by tail_sum_kernel, plus, where a frame is rendered in sample slices, one read and one "
                              "write-through store of the accumulator per slice" % (rnd, w)}
json.dump(pmc_all, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc.json"), "w"), indent=1)
if traffic:
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print("collected", sorted(pmc_all), "traffic", traffic.get("c3", {}).get("hbm_bytes_per_launch"))
