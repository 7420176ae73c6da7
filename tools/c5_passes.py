#!/usr/bin/env python3
"""C5 as BASELINE.json words it (1024 spp of progressive accumulation: 16 passes of 64 spp with RT_RENDER_CONTINUE) at several limits
of the colour buffer of the frame's end; a 64-spp pass of a 4K frame would like 3.2 GB of it.
    python tools/c5_passes.py [MiB ...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
limits = [int(a) for a in sys.argv[1:]] or [2048, 3300, 4096, 0]
wl = bench.WORKLOADS["c5"]
ctx = api.Context(0)
ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
W, H, spp, depth = wl["W"], wl["H"], wl["spp"], wl["depth"]
dev = torch.device("cuda", 0)
acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
per = spp // 16
for mb in limits:
    ctx.set_tail_limit(mb)
    ctx.render_into(W, H, acc, per, depth, seed=0xC0FFEE)    # warm-up: the buffer of this size
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(16):
        ctx.render_into(W, H, acc, per, depth, seed=0xC0FFEE, spp_begin=i * per, flags=api.RENDER_CONTINUE if i else 0)
    torch.cuda.synchronize(dev)
    print(f"c5 as 16 passes of {per} spp, tail limit {mb} MiB: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
ctx.close()
