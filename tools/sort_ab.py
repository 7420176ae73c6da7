#!/usr/bin/env python3
"""Frame time with and without block-sorted shading (rt_set_shading_sort), and a bit-for-bit comparison of the two frames.
    python tools/sort_ab.py [workload[:spp[:WxH]] ...]      e.g. c2 box c3:64 c2:64:1280x720"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
ctx = api.Context(0)
for spec in (sys.argv[1:] or ["c2", "c3:64"]):
    parts = spec.split(":")
    name = parts[0]
    wl = dict(bench.WORKLOADS[name])
    spp = int(parts[1]) if len(parts) > 1 and parts[1] else wl["spp"]
    W, H = wl["W"], wl["H"]
    if len(parts) > 2:
        W, H = (int(v) for v in parts[2].split("x"))
    ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
    acc = ctx.alloc(W * H * 16).zero()
    rays = ctx.alloc(8).zero()
    frames = {}
    for rnd in range(2):
        for mode in (0, 1):
            ctx.set_shading_sort(mode)
            best = 1e9
            for _ in range(3):
                rays.zero()
                ctx.render_into(W, H, acc, spp, int(os.environ.get("RT_AB_DEPTH", wl["depth"])), seed=int(os.environ.get("RT_AB_SEED", "1")), rays=rays,
                                band_first=int(os.environ.get("RT_AB_BAND_FIRST", "0")), band_stride=int(os.environ.get("RT_AB_BAND_STRIDE", "1")))
                best = min(best, ctx.last_kernel_ms()[0])
            n = int(rays.to_numpy("uint64", (1,))[0])
            frames[mode] = acc.to_numpy("uint32", (H, W, 4)).copy()
            print(f"{name} {W}x{H} spp={spp} sort={mode}: {best:.2f} ms, {n / best / 1e3:.0f} Mrays/s, {n} rays", flush=True)
    print(f"{name}: frames identical: {bool(np.array_equal(frames[0], frames[1]))}", flush=True)
    acc.free(); rays.free()
ctx.set_shading_sort(0)
ctx.close()
