#!/usr/bin/env python3
"""Sweep of the tail's size (tiles handed out in sample units) and unit (samples) on the 1080p / 64 spp workloads.
    python tools/tail_sweep.py [workload ...]"""
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
    W, H, spp = wl["W"], wl["H"], wl["spp"]
    acc = ctx.alloc(W * H * 16).zero()
    rays = ctx.alloc(8).zero()
    for tiles in (0, 4096, 8192, 16384, 1 << 30):
        for chunk in (2, 4, 8):
            if tiles == 0 and chunk != 4:
                continue
            os.environ["RT_TAIL_TILES"] = str(tiles)
            os.environ["RT_TAIL_CHUNK"] = str(chunk)
            best = 1e9
            for _ in range(3):
                rays.zero()
                ctx.render_into(W, H, acc, spp, wl["depth"], seed=1, rays=rays)
                best = min(best, ctx.last_kernel_ms()[0])
            n = int(rays.to_numpy("uint64", (1,))[0])
            print(f"{name} {W}x{H} spp={spp} tail tiles {tiles} unit {chunk}: {best:.2f} ms, {n / best / 1e3:.0f} Mrays/s", flush=True)
ctx.close()
