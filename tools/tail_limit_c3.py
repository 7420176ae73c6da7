#!/usr/bin/env python3
"""C3 (4K, 256 spp, 485 spheres) with the product's settings at several limits of the colour buffer of the frame's end
(rt_set_tail_limit): kernel time per frame, min / mean over the frames after a warm-up.
    python tools/tail_limit_c3.py [MiB ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
limits = [int(a) for a in sys.argv[1:]] or [1024, 1280, 1536, 1792, 2048]
wl = bench.WORKLOADS["c3"]
ctx = api.Context(0)
ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
W, H, spp = wl["W"], wl["H"], wl["spp"]
acc = ctx.alloc(W * H * 16).zero()
for rnd in range(2):                    # the limits interleaved twice: drift of the box does not favour one of them
    for mb in limits:
        ctx.set_tail_limit(mb)
        ts = []
        for i in range(6):
            ctx.render_into(W, H, acc, spp, wl["depth"], seed=1)
            if i:
                ts.append(ctx.last_kernel_ms()[0])
        print(f"c3 tail limit {mb:5d} MiB: min {min(ts):.2f} mean {sum(ts) / len(ts):.2f} ms  ({' '.join('%.1f' % t for t in ts)})", flush=True)
ctx.close()
