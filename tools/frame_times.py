#!/usr/bin/env python3
"""Kernel time of every one of n consecutive frames of a workload (run-to-run spread): python tools/frame_times.py [workload] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = api.Context(0)
ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
W, H = wl["W"], wl["H"]
acc = ctx.alloc(W * H * 16).zero()
ts = []
for i in range(n):
    ctx.render_into(W, H, acc, wl["spp"] if i else 8, wl["depth"], seed=0xC0FFEE)
    ts.append(ctx.last_kernel_ms()[0])
print(wl["desc"], " ".join(f"{t:.2f}" for t in ts[1:]))
ctx.close()
