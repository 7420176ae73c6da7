#!/usr/bin/env python3
"""Experiment: the triangle group through the flat chunk / word / batch levels against the per-lane BVH, by mesh size.
    python tools/bvh_crossover.py [n ...]     (n x n height field = 2 (n-1)^2 triangles)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_amd import api, scenes  # noqa: E402

W, H, spp, depth = 1920, 1080, 8, 8
ctx = api.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [36, 71, 142, 200]:
    sc = scenes.tri_mesh_scene(n)
    line = f"{2 * (n - 1) ** 2:7d} triangles:"
    for name, env in (("flat", "1000000000"), ("bvh", "1")):
        os.environ["RT_TRI_BVH_MIN"] = env
        ctx.upload_scene(sc)
        acc = ctx.alloc(W * H * 16).zero(); rays = ctx.alloc(8).zero()
        best = 1e9
        for _ in range(3):
            rays.zero()
            ctx.render_into(W, H, acc, spp, depth, seed=1, rays=rays)
            best = min(best, ctx.last_kernel_ms()[0])
        nr = int(rays.to_numpy("uint64", (1,))[0])
        line += f"  {name} {best:8.2f} ms ({nr / best / 1e3:6.0f} Mrays/s)"
        acc.free(); rays.free()
    print(line, flush=True)
ctx.close()
