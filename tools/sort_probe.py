#!/usr/bin/env python3
"""Ray count and accumulator hash of single renders, alternating rt_set_shading_sort modes (a probe for order-dependent differences)."""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
ctx = api.Context(0)
w, h, n, depth = 1280, 720, 16, 8
for rep in range(3):
    ctx.upload_scene(scenes.get_scene("builtin"))
    for mode in (0, 1, 1, 0, 1):
        ctx.set_shading_sort(mode)
        buf = ctx.alloc(w * h * 16).zero()
        r = ctx.alloc(8).zero()
        ctx.render_into(w, h, buf, n, depth, seed=9, rays=r, band_first=1, band_stride=3)
        a = buf.to_numpy(np.uint32, (h, w, 4))
        print(rep, mode, int(r.to_numpy(np.uint64, (1,))[0]), zlib.crc32(a.tobytes()), flush=True)
        buf.free(); r.free()
ctx.close()
