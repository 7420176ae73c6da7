#!/usr/bin/env python3
"""Where the 6 % of C5's 16 x 64-spp form over one 1024-spp call go: kernel time of every pass, and of single calls of 64 .. 512 spp.
    python tools/c5_pass_times.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
wl = bench.WORKLOADS["c5"]
ctx = api.Context(0)
ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
W, H, spp, depth = wl["W"], wl["H"], wl["spp"], wl["depth"]
dev = torch.device("cuda", 0)
acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
per = spp // 16
ctx.render_into(W, H, acc, per, depth, seed=0xC0FFEE)
for what, flag_of in (("RT_RENDER_CONTINUE", lambda i: api.RENDER_CONTINUE if i else 0), ("plain (every pass overwrites)", lambda i: 0)):
    ts = []
    for i in range(16):
        ctx.render_into(W, H, acc, per, depth, seed=0xC0FFEE, spp_begin=i * per, flags=flag_of(i))
        ts.append(ctx.last_kernel_ms()[0])
    print(f"16 passes of {per} spp, {what}: " + " ".join(f"{t:.1f}" for t in ts) + f"  sum {sum(ts):.1f} ms", flush=True)
for n in (64, 128, 256, 512):
    ctx.render_into(W, H, acc, n, depth, seed=0xC0FFEE)
    ctx.render_into(W, H, acc, n, depth, seed=0xC0FFEE)
    t = ctx.last_kernel_ms()[0]
    print(f"one call of {n} spp: {t:.1f} ms = {t / n:.4f} ms per sample", flush=True)
ctx.close()
