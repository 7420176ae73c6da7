#!/usr/bin/env python3
"""One-off check at full frame sizes: render with RT_RENDER_VERIFY, which runs hitFromList's list-order loop
beside the box / filter / exact scan for EVERY ray on the device and counts differing (primitive, t bits).
    python tools/verify_full.py [workload[:spp] ...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402

ctx = api.Context(0)
for spec in sys.argv[1:] or ["c3:256", "c2:64", "box:64", "c4:4", "c5:2"]:
    name, _, spp = spec.partition(":")
    wl = dict(bench.WORKLOADS[name])
    spp = int(spp) if spp else wl["spp"]
    ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
    W, H = wl["W"], wl["H"]
    acc = ctx.alloc(W * H * 16).zero()
    rays = ctx.alloc(8).zero()
    t0 = time.time()
    ctx.render_into(W, H, acc, spp, wl["depth"], seed=int(os.environ.get('RT_VERIFY_SEED', '0xC0FFEE'), 0), rays=rays, flags=api.RENDER_VERIFY)
    n = int(rays.to_numpy("uint64", (1,))[0])
    print(f"{wl['desc']} at {spp} spp: {n} rays verified against the list-order loop, {ctx.verify_mismatches()} mismatches ({time.time() - t0:.1f} s)", flush=True)
    acc.free(); rays.free()
ctx.close()
