#!/usr/bin/env python3
"""How much of a frame is its tail? The same scene and the same number of samples in total, spread over more or fewer pixels:
a lane's unit of work is one pixel with all its samples, so a frame of few pixels with many samples ends in a long tail.
    python tools/tail_probe.py [workload ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
ctx = api.Context(0)
for name in (sys.argv[1:] or ["c2", "c4", "box"]):
    wl = dict(bench.WORKLOADS[name])
    ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
    for W, H, spp in ((960, 540, 256), (1920, 1080, 64), (3840, 2160, 16), (7680, 4320, 4)):
        acc = ctx.alloc(W * H * 16).zero()
        rays = ctx.alloc(8).zero()
        best = 1e9
        for _ in range(3):
            rays.zero()
            ctx.render_into(W, H, acc, spp, wl["depth"], seed=1, rays=rays)
            best = min(best, ctx.last_kernel_ms()[0])
        n = int(rays.to_numpy("uint64", (1,))[0])
        print(f"{name} {W}x{H} spp={spp}: {best:.2f} ms, {n / best / 1e3:.0f} Mrays/s, {W * H / 64 / 4096:.1f} tiles per wave", flush=True)
        del acc, rays
ctx.close()
