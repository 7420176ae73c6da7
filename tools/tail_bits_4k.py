#!/usr/bin/env python3
"""4K frames of 32 samples rendered with no colour buffer for the frame's end, with the default one and with one that is
smaller than the frame would like: the accumulators must be the same bits (the policy of render_impl decides only HOW the
frame's end is cut into units).   python tools/tail_bits_4k.py"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402
import bench  # noqa: E402
ctx = api.Context(0)
bad = 0
for name in ("c3", "c5"):
    wl = bench.WORKLOADS[name]
    ctx.upload_scene(scenes.get_scene(wl["scene"], **wl["kw"]))
    W, H, spp = wl["W"], wl["H"], 32
    acc = ctx.alloc(W * H * 16)
    ref = None
    for mb, slices in ((0, (0, 16)), (0, (8, 16)), (2048, (8, 16)), (300, (8, 16)), (100, (0, 16))):
        ctx.set_tail_limit(mb); ctx.set_sample_slices(*slices)
        acc.zero()
        ctx.render_into(W, H, acc, spp, wl["depth"], seed=3)
        got = acc.to_numpy(np.uint32, (H, W, 4)).copy()
        if ref is None:
            ref = got
        same = bool(np.array_equal(got, ref))
        bad += not same
        print(f"{name} 4K {spp} spp, tail limit {mb} MiB, slices {slices}: {ctx.last_kernel_ms()[0]:.2f} ms, {'same bits' if same else 'DIFFERENT'}", flush=True)
    acc.free()
ctx.close()
sys.exit(1 if bad else 0)
