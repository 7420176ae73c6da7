#!/usr/bin/env python3
"""rt_set_tile_order(0 / 1) on the workloads: the accumulators must be the same bits; kernel time of the frames (events around all launches
of a call, the measuring pass and the sort included).   python tools/tile_order_ab.py [workload:spp ...]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
cases = sys.argv[1:] or ["c3:256", "c3:32", "c5:64", "c4:64", "c2:64", "box:64"]
ctx = api.Context(0)
bad = 0
for case in cases:
    name, spp = case.split(":")
    spp = int(spp)
    wl = bench.WORKLOADS[name]
    ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
    W, H = wl["W"], wl["H"]
    acc = ctx.alloc(W * H * 16)
    rays = ctx.alloc(8)
    ref = None
    for mode in (0, 1, 2, 0, 1, 2):
        ctx.set_tile_order(mode)
        ts = []
        for i in range(4):
            acc.zero(); rays.zero()
            ctx.render_into(W, H, acc, spp, wl["depth"], seed=1, rays=rays)
            if i:
                ts.append(ctx.last_kernel_ms()[0])
        got = acc.to_numpy(np.uint32, (H, W, 4)).copy()
        n = int(rays.to_numpy("uint64", (1,))[0])
        if ref is None:
            ref = (got, n)
        same = bool(np.array_equal(got, ref[0])) and n == ref[1]
        bad += not same
        print(f"{name} {W}x{H} {spp} spp, tile order {mode}: {' '.join('%.2f' % t for t in ts)} ms ({ctx.last_kernel_ms()[1]} launches), {'same bits and rays' if same else 'DIFFERENT'}", flush=True)
    acc.free(); rays.free()
ctx.set_tile_order(1)
ctx.close()
sys.exit(1 if bad else 0)
