#!/usr/bin/env python3
"""rt_render_pixels on a caller-drawn pixel list (RayView::renderOneRay's schedule, rayview.cpp:69-70): cost of the list's ORDER.
Every pixel of the frame once, in random order, in tile order (what rt_render does), and random but sorted by 8x8 tile.
    python tools/pixels_order.py [workload] [spp]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = api.Context(0)
ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
W, H = wl["W"], wl["H"]
acc = ctx.alloc(W * H * 16).zero()
rays = ctx.alloc(8).zero()
rng = np.random.default_rng(1)
perm = rng.permutation(W * H).astype(np.int32)
y, x = np.divmod(np.arange(W * H, dtype=np.int64), W)
tile_key = (y // 8) * ((W + 7) // 8) + (x // 8)
in_tile = (y % 8) * 8 + (x % 8)
tile_order = np.lexsort((in_tile, tile_key)).astype(np.int32)
lists = {"frame (rt_render)": None, "tile order": tile_order, "random": perm, "random, 65536 at a time sorted by tile": None}
chunks = [np.sort(perm[i:i + 65536]) for i in range(0, perm.size, 65536)]   # a caller's slices, each sorted by pixel index
def key_sorted(a):
    yy, xx = np.divmod(a.astype(np.int64), W)
    return a[np.lexsort(((yy % 8) * 8 + xx % 8, (yy // 8) * ((W + 7) // 8) + xx // 8))]
lists["random, 65536 at a time sorted by tile"] = np.concatenate([key_sorted(c) for c in chunks])
for name, px in lists.items():
    best = 1e9
    for _ in range(3):
        rays.zero()
        if px is None:
            ctx.render_into(W, H, acc, spp, wl["depth"], seed=1, rays=rays)
        else:
            d = ctx.alloc(px.nbytes).from_numpy(px)
            ctx.render_pixels(W, H, acc, d, spp, wl["depth"], seed=1, rays=rays)
            d.free()
        best = min(best, ctx.last_kernel_ms()[0])
    n = int(rays.to_numpy("uint64", (1,))[0])
    print(f"{wl['desc']} spp={spp} {name:45s} {best:8.2f} ms {n / best / 1e3:8.0f} Mrays/s", flush=True)
ctx.close()
