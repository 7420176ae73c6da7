#!/usr/bin/env python3
"""Frame time against the sample-slice settings (rt_set_sample_slices): smallest slice 0 (whole pixels), 4, 8, 16, 32 and slice limits.
    python tools/slice_sweep.py [workload[:spp[:band_stride]] ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
ctx = api.Context(0)
for spec in (sys.argv[1:] or ["c3", "c2", "c4"]):
    parts = spec.split(":")
    name = parts[0]
    wl = dict(bench.WORKLOADS[name])
    spp = int(parts[1]) if len(parts) > 1 and parts[1] else wl["spp"]
    stride = int(parts[2]) if len(parts) > 2 else 1
    ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
    W, H = wl["W"], wl["H"]
    acc = ctx.alloc(W * H * 16).zero()
    rays = ctx.alloc(8).zero()
    for smin, smax, mb in ((0, 16, 0), (8, 16, 0), (8, 16, 256), (8, 16, 1024), (8, 16, 2048), (8, 16, 4096), (0, 16, 1024), (0, 16, 2048)):
        ctx.set_sample_slices(smin, smax)
        ctx.set_tail_limit(mb)
        best = 1e9
        for _ in range(2 if name == "c5" else 3):
            rays.zero()
            ctx.render_into(W, H, acc, spp, wl["depth"], seed=1, rays=rays, band_first=0, band_stride=stride)
            best = min(best, ctx.last_kernel_ms()[0])
        n = int(rays.to_numpy("uint64", (1,))[0])
        print(f"{name} spp={spp} stride={stride} slice_min={smin} tail_limit={mb} MiB: {best:.2f} ms, {n / best / 1e3:.0f} Mrays/s", flush=True)
    acc.free(); rays.free()
ctx.close()
