#!/usr/bin/env python3
"""Sweep of the tail's size (the colour buffer's limit, rt_set_tail_limit) on the 1080p / 64 spp workloads.
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
    # (round 2 swept the number of tail tiles and the unit through environment switches of the library, profiles/r2_tail_sweep.txt;
    # since round 3 the size follows from rt_set_tail_limit and the unit is 8 samples: tools/slice_sweep.py sweeps both mechanisms)
    for mb in (0, 64, 256, 1024, 2048, 4096):
        ctx.set_sample_slices(0, 16)          # whole pixels outside the tail: the tail's effect alone
        ctx.set_tail_limit(mb)
        best = 1e9
        for _ in range(3):
            rays.zero()
            ctx.render_into(W, H, acc, spp, wl["depth"], seed=1, rays=rays)
            best = min(best, ctx.last_kernel_ms()[0])
        n = int(rays.to_numpy("uint64", (1,))[0])
        print(f"{name} {W}x{H} spp={spp} tail limit {mb} MiB: {best:.2f} ms, {n / best / 1e3:.0f} Mrays/s", flush=True)
    ctx.set_sample_slices(8, 16)
    ctx.set_tail_limit(1400)
ctx.close()
