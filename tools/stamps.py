#!/usr/bin/env python3
"""Diagnostic: render a workload with the RT_STAMPS build of the library (tools/libray_amd_stamps.so,
built by hand with -DRT_STAMPS) and print the cycle share of each section of the render loop.
    python tools/stamps.py [workload] [spp] [depth]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stride = int(os.environ.get("RT_STAMPS_BAND_STRIDE", "1"))
if len(sys.argv) > 3:
    wl["depth"] = int(sys.argv[3])
ctx = api.Context(0, lib_path=os.path.join(ROOT, "tools", os.environ.get("RT_STAMPS_LIB", "libray_amd_stamps.so")))
scene = scenes.get_scene(wl["scene"], **wl["kw"])
ctx.upload_scene(scene)
if "RT_SORT" in os.environ:   # block-sorted shading: "filter" is then count + barriers + item hand-over, "shade" the shading of the items, "other" the wait for the results
    ctx.set_shading_sort(int(os.environ["RT_SORT"]))
W, H = wl["W"], wl["H"]
acc = ctx.alloc(W * H * 16).zero()
rays = ctx.alloc(8).zero()
for _ in range(2):
    rays.zero()
    ctx.render_into(W, H, acc, spp, wl["depth"], seed=1, rays=rays, band_first=0, band_stride=stride)
ms, _ = ctx.last_kernel_ms()
out = (C.c_uint64 * 12)()
ctx.lib.raw().rt_debug_stamps.argtypes = [C.POINTER(C.c_uint64)]
ctx.lib.raw().rt_debug_stamps(out)
t = list(out)
n_rays = int(rays.to_numpy("uint64", (1,))[0])
names = ["assign+new sample", "exact group", "filter", "candidate walk", "rect+tri(+fallback)", "shade", "unwind+store", "other"]
batches = t[1]
t[1] = 0
other = t[7]
t[7] = 0
tot = sum(t[:8])
print(f"{wl['desc']} at {spp} spp, depth {wl['depth']}: {ms:.2f} ms, {n_rays / ms / 1e3:.0f} Mrays/s (stamped build; read shares, not speed)")
scans = t[11]
for n, v in zip(names, t[:8]):
    print(f"  {n:22s} {100.0 * v / tot:6.2f} %   {v / max(scans, 1):9.0f} cycles per wave-scan")
print(f"  slot 1 (triangle scenes: cycles in level 0 of the filter; else hand-outs refused with four unfinished units) {batches} ({batches / max(scans, 1):.1f} per wave-scan); slot 7 (events) {other} ({other / max(scans, 1):.4f} per wave-scan)")
print(f"  wave-scans {scans}, rays per wave-scan {n_rays / max(scans, 1):.1f} (64 = all lanes busy)")
print(f"  walk passes per wave-scan {t[8] / max(scans, 1):.2f}; per ray: flagged sphere pairs {t[9] / max(n_rays, 1):.2f}, sphere candidates (d >= 0) {t[10] / max(n_rays, 1):.2f}")
ctx.close()
