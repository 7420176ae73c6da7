#!/usr/bin/env python3
"""Diagnostic: when do the waves of a frame end? Renders with a stamped build (RT_STAMPS) and prints the distribution of the waves'
end times relative to the first start, as fractions of the frame.   python tools/wave_times.py <lib under tools/> [workload] [spp]"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
lib = sys.argv[1]
wl = dict(bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "c4"])
spp = int(sys.argv[3]) if len(sys.argv) > 3 else wl["spp"]
ctx = api.Context(0, lib_path=os.path.join(ROOT, "tools", lib))
ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
W, H = wl["W"], wl["H"]
acc = ctx.alloc(W * H * 16).zero()
for _ in range(2):
    ctx.render_into(W, H, acc, spp, wl["depth"], seed=1)
ms = ctx.last_kernel_ms()[0]
n = 4096
t = np.zeros((n, 2), np.uint64)
ctx.lib.raw().rt_debug_wave_times.argtypes = [C.c_void_p, C.c_int]
ctx.lib.raw().rt_debug_wave_times(t.ctypes.data, n)
t = t.astype(np.float64)
life = (t[:, 1] - t[:, 0])          # (per wave: the counters of different XCDs do not share an origin)
frame = life.max()
print(f"{lib} {wl['desc']} spp={spp}: {ms:.2f} ms; wave lifetime mean {life.mean() / frame:.3f} of the longest")
print("  lifetime quantiles (fraction of the longest):", " ".join(f"{q}%={np.percentile(life, q) / frame:.3f}" for q in (0, 1, 10, 25, 50, 75, 90, 99)))
ctx.close()
