#!/usr/bin/env python3
"""Experiment builds must still be exact: render with RT_RENDER_VERIFY through an alternative library and time it.
    python tools/exp_verify.py <lib.so> [workload] [spp]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
lib = sys.argv[1]
wl = dict(bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "c3"])
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = api.Context(0, lib_path=os.path.join(ROOT, lib))
ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
W, H = wl["W"], wl["H"]
acc = ctx.alloc(W * H * 16).zero(); rays = ctx.alloc(8).zero()
ctx.render_into(W, H, acc, 4, wl["depth"], seed=1, rays=rays, flags=api.RENDER_VERIFY)
print(f"{lib} {wl['desc']}: verify 4 spp: {int(rays.to_numpy('uint64', (1,))[0])} rays, {ctx.verify_mismatches()} mismatches")
best = 1e9
for _ in range(3):
    rays.zero()
    ctx.render_into(W, H, acc, spp, wl["depth"], seed=1, rays=rays)
    best = min(best, ctx.last_kernel_ms()[0])
n = int(rays.to_numpy("uint64", (1,))[0])
print(f"{lib} {wl['desc']} spp={spp}: {best:.2f} ms, {n / best / 1e3:.0f} Mrays/s")
ctx.close()
